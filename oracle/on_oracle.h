/*
 * on_oracle.h -- CPU restatement of the reference's O(N) scalar path (the "O(1) radial" Metropolis
 * update in the .cl files of O(N)/oncl and its measurements).
 *
 * TEST INFRASTRUCTURE ONLY.  This is the checker the HIP kernels are compared against; nothing in
 * the product library may include, link or call it.
 *
 * Parity status:
 *   measurements (action, field, field^2, correlators): PINNED in double precision -- checked
 *     against the reference's own CPU re-measurement (O(N)/suncl/suncpu.cpp:93-270, compiled from
 *     its sources into oracle/_ref/qcdgpu_on_check by `make ref_on`) through tests/golden/on_*.json;
 *   update pass, hot start, single precision: "parity unpinned" -- the reference has them as
 *     OpenCL kernels only (no device here); restated from O(N)/oncl/oncl.cl:49-366 and
 *     O(N)/oncl/o1cl.cl:41-160 with the arithmetic types C gives the same expressions
 *     (unsuffixed literals are double, so most of the "single precision" action is evaluated in
 *     double and rounded to float where the kernel stores into an hgpu_float).
 */
#ifndef ON_ORACLE_H
#define ON_ORACLE_H

#include "qcd_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int L[4];              /* Nx, Ny, Nz, Nt; site index gid = y + z*Ny + t*Ny*Nz + x*Ny*Nz*Nt */
    int sites;
    int precision;         /* 0: float field and parameters, 1: double */
    int nhit;
    int skgroup;           /* SKGROUP: N of the O(N) group in the action */
    double z, lambda, zeta, eta, b;
    double sqrt_z_lambda_zeta, max_U;   /* derived, O(N)/suncl/suncl.cpp:1244-1253 */
    void *field;           /* float[sites] or double[sites] */
    int nskip;
    qo_ranlux *streams;    /* sites/2 streams, one per thread of an update pass */
    unsigned long long rng_draws;
    double acc[2];         /* ACC_RATE sums of the last even / odd pass (oncl.cl:86-94,216-218) */
} qon_lattice;

qon_lattice *qon_create(const int L[4], int precision, int nhit, int skgroup, double z, double lambda,
                        double zeta, double eta, double b, int lux_level, unsigned randseries);
void qon_destroy(qon_lattice *lat);
void qon_init(qon_lattice *lat, int hot);            /* oncl.cl:49-72 */
void qon_update_pass(qon_lattice *lat, int parity);  /* 0: update_even, 1: update_odd, oncl.cl:74-366 */
void qon_sweep(qon_lattice *lat);                    /* odd then even, O(N)/suncl/suncl.cpp:2228-2232 */
double qon_measure_action(const qon_lattice *lat);   /* on_measurements_cl.cl:48-107, per site */
void qon_measure_field(const qon_lattice *lat, double out[2]);                       /* :122-154 */
void qon_measure_correlators(const qon_lattice *lat, const int step[4], double out[2]); /* :167-214 */
double qon_get(const qon_lattice *lat, int gid);
void qon_set(qon_lattice *lat, int gid, double v);

#ifdef __cplusplus
}
#endif
#endif
