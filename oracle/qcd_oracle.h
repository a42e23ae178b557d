/*
 * qcd_oracle.h -- CPU restatement of the QCDGPU sequential SU(N) path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the checker the HIP kernels are compared
 * against; nothing in the product library may include, link or call it.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose arithmetic it restates.  Parity status: PINNED -- the restatement is
 * checked against the compiled reference (oracle/_ref/qcdgpu_cpu, built from
 * the reference's own sources by oracle/Makefile) through the fixtures under
 * tests/golden/, in the sequential-stream ("seq") and deterministic ("gid")
 * modes.  The Polyakov loop, the Wilson loop and the F_mu_nu traces are pinned
 * against the reference's CPU re-measurement of a configuration
 * (SU(N)/suncl/su3cpu.cpp:1616-1704, compiled into oracle/_ref/qcdgpu_sun_check by
 * `make ref_sun`; fixtures tests/golden/chk_su3_*, chk_su2_*);
 * the hot start, the overrelaxation and the plane profiles are "parity unpinned".
 */
#ifndef QCD_ORACLE_H
#define QCD_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { double re, im; } qo_cplx;

/* SU(N)/suncpp/su3/algebra_su3.h: full 3x3 complex matrix, rows u, v, w */
typedef struct { qo_cplx u1, u2, u3, v1, v2, v3, w1, w2, w3; } qo_su3;
/* SU(N)/suncpp/su2/algebra_su2.h: full 2x2 complex matrix, rows u, v */
typedef struct { qo_cplx u1, u2, v1, v2; } qo_su2;

/* RANLUX, SU(N)/random/random.cpp:504-574 */
typedef struct {
    float seeds[24];
    int   i24, j24, in24;
    float carry;
    int   nskip;
} qo_ranlux;

enum { QO_RNG_SEQ = 0, QO_RNG_GID = 1, QO_RNG_PERSITE = 2 };

typedef struct {
    int group;            /* 2 or 3 */
    int nd;               /* 4 */
    int L[4];             /* Nx, Ny, Nz, Nt */
    int sites;
    int nhit;
    double beta;
    int ints;             /* 0 hot (per-thread streams only; the reference's CPU path has none), 1 cold, 2 gid */
    int rng_mode;         /* QO_RNG_* */
    int nskip;            /* luxury skip: 24/48/97/223/389 */
    unsigned randseries;
    void *links;          /* qo_su3[nd*sites] or qo_su2[nd*sites], index gid*nd+dir */
    qo_ranlux seq;        /* the one sequential stream */
    qo_ranlux *streams;   /* sites/2/share streams in QO_RNG_PERSITE mode */
    unsigned long long rng_draws; /* numbers consumed (all modes but gid) */
    /* twisted boundary condition (device path and CPU re-measurement of the reference only; its sequential CPU
     * program has none): see qo_set_twist */
    double phi, omega;
    void *tw_backup;      /* the stored U_y links of the slice x = Nx-1 while the table holds their twisted form */
    int tw_depth;
    int share;            /* sites of one parity per stream in QO_RNG_PERSITE mode (qo_set_rng_share), 1 by default */
} qo_lattice;

/* glibc rand() restated (TYPE_3 additive feedback), so that seeds do not depend on the libc */
typedef struct { int32_t r[34]; int idx; uint32_t ring[31]; int f, b; } qo_glibc_rand;
void     qo_srand(qo_glibc_rand *g, unsigned seed);
int      qo_rand(qo_glibc_rand *g);

void  qo_ranlux_init(qo_ranlux *r, int jseed, int nskip);
float qo_ranlux_next(qo_ranlux *r);
void  qo_ranlux_align(qo_ranlux *r);           /* drop the rest of the delivered group of 24 */
int   qo_nskip_for_level(int level);          /* RANLUX0..4 -> 24,48,97,223,389 */
int   qo_stream_jseed(unsigned randseries, int stream); /* (stream+1)-th rand() after srand */

/* index maps, SU(N)/suncpp/coord_work/coord_work.cpp */
void     qo_gid_to_coords(const int L[4], unsigned gid, int c[4]);
unsigned qo_coords_to_gid(const int L[4], const int c[4]);
int      qo_neighbour(const int L[4], int gid, int dir);
int      qo_neighbour_back(const int L[4], int gid, int dir);
unsigned qo_odd_gid(const int L[4], unsigned i);
unsigned qo_even_gid(const int L[4], unsigned i);

/* algebra */
qo_su3 qo_mul3(const qo_su3 *a, const qo_su3 *b);
qo_su3 qo_herm3(const qo_su3 *a);
void   qo_gram_schmidt3(qo_su3 *a);
void   qo_matrix_gid3(qo_su3 *a, int gid, int dir, int sites);
qo_su2 qo_mul2(const qo_su2 *a, const qo_su2 *b);
qo_su2 qo_herm2(const qo_su2 *a);
void   qo_matrix_gid2(qo_su2 *a, int gid, int dir, int sites);
/* hot-start links from 12 / 4 uniform numbers (device path of the reference only; "parity unpinned") */
void   qo_hot_link3(qo_su3 *m, const float r[12]);
void   qo_hot_link2(qo_su2 *m, const float r[4]);

/* Twisted boundary condition, PHI / OMEGA of the reference (-D TBC, SU(N)/suncl/suncl.cpp:3286).  Every read of U_y at
 * x = Nx-1 by an update pass or a measurement returns Omega U_y: SU(3) Omega = diag(e^{i(phi+omega)/2}, e^{i(omega-phi)/2},
 * e^{-i omega}) (SU(N)/suncl/su3_matrix_memory.cl:57-98, CPU re-measurement SU(N)/suncl/su3cpu.cpp:165-197), SU(2) Omega =
 * diag(e^{i phi/2}, e^{-i phi/2}) (su2_matrix_memory.cl:63-88, su2cpu.cpp:164-183).  An updated link is stored as the heat-bath
 * returns it, without taking the twist off (SU(N)/suncl/suncl.cl:525-536).  Measurements: pinned by the reference's CPU
 * re-measurement; update pass: device only in the reference, "parity unpinned". */
void qo_set_twist(qo_lattice *lat, double phi, double omega);
/* per-thread streams shared by `share` sites of one parity (Nt/share apart in t); before any number is drawn; 0 = ok */
int  qo_set_rng_share(qo_lattice *lat, int share);

/* lattice life cycle */
qo_lattice *qo_create(int group, const int L[4], double beta, int nhit, int ints,
                      int rng_mode, int lux_level, unsigned randseries);
void qo_destroy(qo_lattice *lat);
void qo_init_links(qo_lattice *lat);

/* update, SU(N)/suncpp/Update/sun_update.h */
void qo_staple3(const qo_lattice *lat, int gid, int dir, qo_su3 *out);
void qo_staple2(const qo_lattice *lat, int gid, int dir, qo_su2 *out);
void qo_update_pass(qo_lattice *lat, int parity /*0 even, 1 odd*/, int dir);
void qo_sweep(qo_lattice *lat);   /* odd X,Y,Z,T then even X,Y,Z,T */
/* overrelaxation: not in the reference ("parity unpinned"); checked by the invariance of the action */
void qo_overrelax_link3(qo_su3 *U, const qo_su3 *stap);
void qo_overrelax_link2(qo_su2 *U, const qo_su2 *stap);
void qo_overrelax_pass(qo_lattice *lat, int parity, int dir);
void qo_overrelax_sweep(qo_lattice *lat);
/* slab variants (x in [x_begin, x_end)) for the multi-rank tests */
void qo_update_pass_range(qo_lattice *lat, int parity, int dir, int x_begin, int x_end);
void qo_plaquette_sums_range(const qo_lattice *lat, int x_begin, int x_end, double out[2]);
void qo_polyakov_sums_range(const qo_lattice *lat, int x_begin, int x_end, double out[4]);

/* one link update with an explicit random source (unit tests) */
void qo_update_link3(qo_su3 *U, const qo_su3 *stap, double beta_eff, int nhit, int gid,
                     int gid_start, int fsites, qo_ranlux *rng, unsigned long long *draws);
void qo_update_link2(qo_su2 *a, const qo_su2 *stap, double beta_eff, int nhit, int gid,
                     int gid_start, int fsites, int final, qo_ranlux *rng,
                     unsigned long long *draws);

/* measurements: out[0]=spat, out[1]=temp, out[2]=total */
void qo_plaquette(const qo_lattice *lat, double out[3]);
void qo_action(const qo_lattice *lat, double out[3]);
/* Wilson loop W(R,T) average (device path of the reference only: "parity unpinned") */
double qo_wilson_loop(const qo_lattice *lat, int R, int T);
/* out = Re P, Im P, |P|^2, |P|^4 averages; "parity unpinned" */
void qo_polyakov(const qo_lattice *lat, double out[4]);

/* profiles along a space axis (0 x, 1 y, 2 z): out[2*i], out[2*i+1] = (spat, temp) plaquette sums
 * resp. (Re, Im) Polyakov-loop sums of the sites with coordinate i; device path of the reference
 * only: "parity unpinned" */
void qo_plaquette_profile_sums(const qo_lattice *lat, int axis, double *out);
void qo_polyakov_profile_sums(const qo_lattice *lat, int axis, double *out);

/* F_mu_nu traces summed over the sites (Gell-Mann matrices 1..8 for SU(3), Pauli 1..3 for SU(2)): field 0 = planes xy,xz,yz, 1 = xt,yt,zt; out = for
 * (index1, index2) x three planes x (re, im); SU(N)/suncl/su3cpu.cpp:1212-1381 */
void qo_fmunu_sums(const qo_lattice *lat, int field, int index1, int index2, double out[12]);

/* statistics, SU(N)/suncpp/Measurements/analysis_cpp.cpp:81-93 */
void qo_estimates(const double *data, int n, double *mean, double *variance);

/* whole run, SU(N)/suncpp/suncpp.h:73-165: fills plq[3*iter], act[3*iter] */
void qo_simulate(qo_lattice *lat, int nav, int iter, int niter, double *plq, double *act);

/* link table in the reference's device layout (double precision):
 * SU(3): 12 rows of double4, row = quad*4+dir; SU(2): 4 rows, row = dir
 * (SU(N)/suncl/su3_matrix_memory.cl:42-176, su2_matrix_memory.cl:55-142) */
void qo_export_device_layout(const qo_lattice *lat, double *table, int rowsize);
void qo_import_device_layout(qo_lattice *lat, const double *table, int rowsize);

#ifdef __cplusplus
}
#endif
#endif
