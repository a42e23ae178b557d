/*
 * on_oracle.c -- CPU restatement of the reference's O(N) scalar path.  TEST INFRASTRUCTURE ONLY
 * (see on_oracle.h for what is pinned against the compiled reference and what is not).
 */
#include "on_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

#define REAL float
#define FN(name) CAT(name, _f32)
#define LOGR logf
#define EXPR expf
#include "on_oracle_body.inc"
#undef REAL
#undef FN
#undef LOGR
#undef EXPR

#define REAL double
#define FN(name) CAT(name, _f64)
#define LOGR log
#define EXPR exp
#include "on_oracle_body.inc"
#undef REAL
#undef FN
#undef LOGR
#undef EXPR

qon_lattice *qon_create(const int L[4], int precision, int nhit, int skgroup, double z, double lambda,
                        double zeta, double eta, double b, int lux_level, unsigned randseries)
{
    qon_lattice *lat = (qon_lattice *)calloc(1, sizeof(*lat));
    memcpy(lat->L, L, sizeof(lat->L));
    lat->sites = L[0] * L[1] * L[2] * L[3];
    lat->precision = precision;
    lat->nhit = nhit;
    lat->skgroup = skgroup;
    lat->z = z;
    lat->lambda = lambda;
    lat->zeta = zeta;
    lat->eta = eta;
    lat->b = b;
    /* model::model_lattice_init, O(N)/suncl/suncl.cpp:1244-1253 */
    lat->sqrt_z_lambda_zeta = 0.25 * sqrt(z / (lambda * zeta));
    if (eta > 0.0) {
        double maxU1 = 1.0 - exp((1.0 - sqrt(1.0 + 4.0 * sqrt(2.0) * eta / sqrt(b))) / eta);
        double maxU2 = 1.0 - exp((1.0 + sqrt(1.0 + 4.0 * sqrt(2.0) * eta / sqrt(b))) / eta);
        if (maxU1 < 0.0) maxU1 = maxU2;
        lat->max_U = maxU1;
    } else {
        lat->max_U = 1.0 - exp(-1.9999999999 / sqrt(b));
    }
    lat->field = calloc((size_t)lat->sites, precision ? sizeof(double) : sizeof(float));
    lat->nskip = qo_nskip_for_level(lux_level);
    /* one stream per thread of an update pass, seeded like the instances of the reference's
     * device generator (SU(N)/random/random.cpp:174-182, 462-465) */
    qo_glibc_rand g;
    qo_srand(&g, randseries);
    const int ns = lat->sites / 2;
    lat->streams = (qo_ranlux *)calloc((size_t)ns, sizeof(qo_ranlux));
    qo_ranlux_init(&lat->streams[0], qo_rand(&g) % 2147483647, lat->nskip);
    for (int i = 1; i < ns; i++) qo_ranlux_init(&lat->streams[i], qo_rand(&g), lat->nskip);
    return lat;
}

void qon_destroy(qon_lattice *lat)
{
    if (!lat) return;
    free(lat->field);
    free(lat->streams);
    free(lat);
}

void qon_init(qon_lattice *lat, int hot)
{
    if (lat->precision) init_f64(lat, hot);
    else init_f32(lat, hot);
}

void qon_update_pass(qon_lattice *lat, int parity)
{
    if (lat->precision) update_pass_f64(lat, parity);
    else update_pass_f32(lat, parity);
}

void qon_sweep(qon_lattice *lat)
{
    qon_update_pass(lat, 1);
    qon_update_pass(lat, 0);
}

double qon_measure_action(const qon_lattice *lat)
{
    return lat->precision ? measure_action_f64(lat) : measure_action_f32(lat);
}

void qon_measure_field(const qon_lattice *lat, double out[2])
{
    if (lat->precision) measure_field_f64(lat, out);
    else measure_field_f32(lat, out);
}

void qon_measure_correlators(const qon_lattice *lat, const int step[4], double out[2])
{
    if (lat->precision) measure_correlators_f64(lat, step, out);
    else measure_correlators_f32(lat, step, out);
}

double qon_get(const qon_lattice *lat, int gid)
{
    return lat->precision ? ((const double *)lat->field)[gid] : (double)((const float *)lat->field)[gid];
}

void qon_set(qon_lattice *lat, int gid, double v)
{
    if (lat->precision) ((double *)lat->field)[gid] = v;
    else ((float *)lat->field)[gid] = (float)v;
}
