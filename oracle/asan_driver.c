/*
 * asan_driver.c -- TEST INFRASTRUCTURE ONLY.  Walks through every entry point of the CPU checker
 * (qcd_oracle.c, on_oracle.c) on small lattices; built by `make asan` with AddressSanitizer and
 * UndefinedBehaviorSanitizer (host code only), so that an out-of-bounds index, a use after free or
 * an overflowing shift in the checker ends the run with a report.  Prints one line per leg.
 */
#include "on_oracle.h"
#include "qcd_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

static void leg(int group, const int L[4], double beta, int nhit, int ints, int mode, int lux, int share, double phi)
{
    qo_lattice *lat = qo_create(group, L, beta, nhit, ints, mode, lux, 54u);
    if (share > 1 && qo_set_rng_share(lat, share)) { printf("share %d refused\n", share); qo_destroy(lat); return; }
    qo_init_links(lat);
    if (phi != 0.0) qo_set_twist(lat, phi, -0.5 * phi);
    double p[3], a[3], pol[4], f[12];
    qo_sweep(lat);
    qo_update_pass_range(lat, 1, 2, 0, L[0] / 2);
    qo_update_pass_range(lat, 1, 2, L[0] / 2, L[0]);
    if (phi == 0.0) qo_overrelax_sweep(lat);
    qo_plaquette(lat, p);
    qo_action(lat, a);
    qo_polyakov(lat, pol);
    const double w = qo_wilson_loop(lat, 2, 2);
    qo_fmunu_sums(lat, 0, 3, 8, f);
    qo_fmunu_sums(lat, 1, group == 3 ? 1 : 2, group == 3 ? 6 : 3, f);
    double sums[2], psums[4];
    qo_plaquette_sums_range(lat, 1, L[0], sums);
    qo_polyakov_sums_range(lat, 0, L[0] - 1, psums);
    for (int axis = 0; axis < 3; axis++) {
        double *prof = (double *)malloc(sizeof(double) * 2 * (size_t)L[axis]);
        qo_plaquette_profile_sums(lat, axis, prof);
        qo_polyakov_profile_sums(lat, axis, prof);
        free(prof);
    }
    const int sites = L[0] * L[1] * L[2] * L[3], rowsize = (1 + (sites - 1) / 256) * 256;
    double *table = (double *)calloc((size_t)(group == 3 ? 12 : 4) * rowsize * 4, sizeof(double));
    qo_export_device_layout(lat, table, rowsize);
    qo_import_device_layout(lat, table, rowsize);
    free(table);
    printf("SU(%d) %dx%dx%dx%d ints %d mode %d lux %d share %d phi %.1f: Plq %.12f S %.12f P %.6f W %.6f draws %llu\n", group, L[0], L[1], L[2],
           L[3], ints, mode, lux, share, phi, p[2], a[2], pol[0], w, lat->rng_draws);
    qo_destroy(lat);
}

int main(void)
{
    static const int A[4] = {4, 4, 4, 4}, B[4] = {4, 6, 2, 8}, C[4] = {5, 4, 3, 7}, D[4] = {2, 2, 2, 2};
    for (int group = 2; group <= 3; group++) {
        leg(group, A, group == 3 ? 6.0 : 2.5, 10, 1, QO_RNG_SEQ, 3, 1, 0.0);
        leg(group, B, 1.0, 12, 1, QO_RNG_PERSITE, 0, 1, 0.0);      /* low beta: long trial loops */
        leg(group, B, 5.5, 3, 0, QO_RNG_PERSITE, 4, 4, 0.0);       /* hot start, shared streams */
        leg(group, B, 5.5, 3, 1, QO_RNG_PERSITE, 2, 2, 0.0);
        leg(group, C, 5.5, 4, 2, QO_RNG_GID, 3, 1, 0.0);           /* odd extents, gid start */
        leg(group, C, 5.5, 4, 1, QO_RNG_SEQ, 1, 1, 0.0);
        leg(group, A, 5.8, 4, 2, QO_RNG_GID, 3, 1, 0.7);           /* twisted boundary condition */
        leg(group, D, 5.8, 2, 1, QO_RNG_PERSITE, 3, 1, 0.0);       /* the smallest lattice: every neighbour wraps */
    }
    /* whole-run driver and statistics */
    {
        double plq[3 * 4], act[3 * 4], mean, var;
        qo_lattice *lat = qo_create(3, A, 6.0, 10, 1, QO_RNG_SEQ, 3, 54u);
        qo_simulate(lat, 1, 4, 2, plq, act);
        qo_estimates(plq, 4, &mean, &var);
        printf("simulate: mean %.12f variance %.3e\n", mean, var);
        qo_destroy(lat);
    }
    /* O(N) scalar model, both precisions */
    for (int prec = 0; prec <= 1; prec++) {
        static const int L[4] = {4, 4, 6, 2}, step[4] = {1, 0, 0, 1};
        qon_lattice *on = qon_create(L, prec, 11, 4, 1.0, 0.5, 1.0, 0.1, 0.2, 3, 7u);
        qon_init(on, 1);
        qon_sweep(on);
        qon_sweep(on);
        double fld[2], cor[2];
        qon_measure_field(on, fld);
        qon_measure_correlators(on, step, cor);
        printf("O(N) precision %d: S %.10f field %.10f corr %.10f draws %llu\n", prec, qon_measure_action(on), fld[0], cor[0], on->rng_draws);
        qon_destroy(on);
    }
    return 0;
}
