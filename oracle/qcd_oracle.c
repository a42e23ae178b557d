/*
 * qcd_oracle.c -- CPU restatement of the QCDGPU sequential SU(N) path.
 *
 * TEST INFRASTRUCTURE ONLY (see qcd_oracle.h).  Plain C, one thread, compiled
 * with -ffp-contract=off so that every product and sum rounds exactly as in the
 * reference build (g++ -O2 on x86-64 emits no fused multiply-adds).  The order
 * of the floating-point operations inside each expression follows the cited
 * reference lines term by term; that is what makes the sequential-stream and
 * deterministic modes reproduce the compiled reference to the last printed
 * digit.
 */
#include "qcd_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* SU(N)/kernel/complex.h:200-202 */
#define QO_PI2 6.2831853071795864769252867665590

/* ------------------------------------------------------------------------- */
/* glibc rand(): stdlib/random_r.c TYPE_3 generator (degree 31, separation 3),
 * glibc 2.35 as shipped in this image.  The reference seeds everything through
 * srand()/rand() (SU(N)/random/random.cpp:174-182, 508, 462-465).            */

void qo_srand(qo_glibc_rand *g, unsigned seed)
{
    if (seed == 0) seed = 1;
    int32_t *r = g->r;
    r[0] = (int32_t)seed;
    for (int i = 1; i < 31; i++) {
        /* 16807 * r[i-1] % 2147483647 without overflow (Schrage) */
        long hi = r[i - 1] / 127773;
        long lo = r[i - 1] % 127773;
        long word = 16807 * lo - 2836 * hi;
        if (word < 0) word += 2147483647;
        r[i] = (int32_t)word;
    }
    for (int i = 0; i < 31; i++) g->ring[i] = (uint32_t)r[i];
    g->f = 3;  /* front pointer: separation 3 */
    g->b = 0;  /* rear pointer */
    for (int i = 0; i < 310; i++) (void)qo_rand(g);
}

int qo_rand(qo_glibc_rand *g)
{
    g->ring[g->f] += g->ring[g->b];
    uint32_t result = g->ring[g->f] >> 1;
    g->f = (g->f + 1) % 31;
    g->b = (g->b + 1) % 31;
    return (int)result;
}

/* ------------------------------------------------------------------------- */
/* RANLUX: SU(N)/random/random.cpp:504-574, constants random.h:100-103        */

#define RL_ICONS   2147483563
#define RL_ITWO24  16777216
#define RL_TWOM24  0.000000059604644775390625f
#define RL_TWOM12  0.000244140625f

int qo_nskip_for_level(int level)
{
    /* SU(N)/random/random.cpp:192-196 */
    static const int t[5] = {24, 48, 97, 223, 389};
    if (level < 0 || level > 4) return 223;
    return t[level];
}

void qo_ranlux_init(qo_ranlux *r, int jseed, int nskip)
{
    /* constructor defaults random.cpp:65-71, then RL_initialize_CPU :504-522 */
    r->nskip = nskip;
    r->i24 = 23;
    r->j24 = 9;
    r->in24 = 0;
    r->carry = 0.0f;
    int js = jseed;
    for (int i = 0; i < 24; i++) {
        int k = js / 53668;
        js = 40014 * (js - k * 53668) - k * 12211;
        if (js < 0) js = js + RL_ICONS;
        r->seeds[i] = ((float)(js % RL_ITWO24)) * RL_TWOM24;
    }
    if (r->seeds[23] == 0) r->carry = RL_TWOM24;
}

static inline float rl_step(qo_ranlux *r)
{
    float uni = r->seeds[r->j24] - r->seeds[r->i24] - r->carry;
    if (uni < 0.0f) {
        uni = uni + 1.0f;
        r->carry = RL_TWOM24;
    } else {
        r->carry = 0.0f;
    }
    r->seeds[r->i24] = uni;
    if (r->i24 == 0) r->i24 = 24;
    if (r->j24 == 0) r->j24 = 24;
    r->i24--;
    r->j24--;
    return uni;
}

float qo_ranlux_next(qo_ranlux *r)
{
    /* RL_produce_one_CPU, random.cpp:523-574 */
    float uni;
    if (r->in24 == 0) {
        r->in24 = 24;
        for (int isk = 0; isk < r->nskip; isk++) (void)rl_step(r);
    }
    uni = rl_step(r);
    if (uni < RL_TWOM12) {
        uni = uni + RL_TWOM24 * r->seeds[r->j24];
        if (uni == 0.0f) uni = RL_TWOM24 * RL_TWOM24;
    }
    r->in24--;
    return uni;
}

void qo_ranlux_align(qo_ranlux *r)
{
    /* Per-thread streams only: drop what is left of the delivered group of 24, so that the next
     * value is the first of a fresh group.  The reference's device path has the same rule in
     * another form: every update kernel starts reading at the front of a freshly produced
     * buffer (SU(N)/suncl/su3_update_cl.cl:679, indprng = GID; PRNG0->produce() in front of
     * every pass, SU(N)/suncl/suncl.cpp:5357-5377) and the numbers a pass did not use are lost. */
    while (r->in24 > 0) {
        (void)rl_step(r);
        r->in24--;
    }
}

int qo_stream_jseed(unsigned randseries, int stream)
{
    /* stream 0: RL_jseed = rand() % 2147483647 (random.cpp:508); stream i>0 takes the next
     * rand() values in order, as the per-instance seed table does (random.cpp:462-465). */
    qo_glibc_rand g;
    qo_srand(&g, randseries);
    int v = qo_rand(&g) % 2147483647;
    for (int i = 0; i < stream; i++) v = qo_rand(&g);
    return v;
}

/* ------------------------------------------------------------------------- */
/* index maps: SU(N)/suncpp/coord_work/coord_work.cpp                          */

void qo_gid_to_coords(const int L[4], unsigned gid, int c[4])
{
    /* :40-57 -- gid = y + z*Ny + t*Ny*Nz + x*Ny*Nz*Nt */
    unsigned z1, z2, z3, z4;
    z4 = gid / (unsigned)(L[1] * L[2] * L[3]);
    z1 = gid - z4 * (unsigned)(L[1] * L[2] * L[3]);
    z3 = z1 / (unsigned)(L[1] * L[2]);
    z1 = z1 - z3 * (unsigned)(L[1] * L[2]);
    z2 = z1 / (unsigned)L[1];
    z1 = z1 - z2 * (unsigned)L[1];
    c[0] = (int)z4; c[1] = (int)z1; c[2] = (int)z2; c[3] = (int)z3;
}

unsigned qo_coords_to_gid(const int L[4], const int c[4])
{
    /* :59-61 */
    return (unsigned)(c[1] + c[2] * L[1] + c[3] * L[1] * L[2] + c[0] * L[1] * L[2] * L[3]);
}

int qo_neighbour(const int L[4], int gid, int dir)
{
    /* :85-105 */
    int c[4];
    qo_gid_to_coords(L, (unsigned)gid, c);
    c[dir]++;
    if (c[dir] >= L[dir]) c[dir] = 0;
    return (int)qo_coords_to_gid(L, c);
}

int qo_neighbour_back(const int L[4], int gid, int dir)
{
    /* :129-149 */
    int c[4];
    qo_gid_to_coords(L, (unsigned)gid, c);
    if (c[dir] <= 0) c[dir] = L[dir];
    c[dir]--;
    return (int)qo_coords_to_gid(L, c);
}

unsigned qo_odd_gid(const int L[4], unsigned i)
{
    /* :151-161 */
    unsigned gde = 2 * i + 1;
    int c[4];
    qo_gid_to_coords(L, gde, c);
    unsigned even_check = (unsigned)(((c[0] + c[1] + c[2] + c[3]) & 1) ^ 1);
    return gde - even_check;
}

unsigned qo_even_gid(const int L[4], unsigned i)
{
    /* :163-175 */
    unsigned gde = 2 * i;
    int c[4];
    qo_gid_to_coords(L, gde, c);
    unsigned odd_check = (unsigned)((c[0] + c[1] + c[2] + c[3]) & 1);
    return gde + odd_check;
}

/* ------------------------------------------------------------------------- */
/* SU(3) algebra: SU(N)/suncpp/su3/algebra_su3.cpp                             */

/* one entry of a 3x3 product, term order of algebra_su3.cpp:121-122 */
static inline qo_cplx dot3(qo_cplx a1, qo_cplx a2, qo_cplx a3, qo_cplx b1, qo_cplx b2, qo_cplx b3)
{
    qo_cplx c;
    c.re = -a1.im * b1.im + a1.re * b1.re - a2.im * b2.im + a2.re * b2.re - a3.im * b3.im + a3.re * b3.re;
    c.im =  a1.re * b1.im + a1.im * b1.re + a2.re * b2.im + a2.im * b2.re + a3.re * b3.im + a3.im * b3.re;
    return c;
}

qo_su3 qo_mul3(const qo_su3 *a, const qo_su3 *b)
{
    /* :118-143 */
    qo_su3 c;
    c.u1 = dot3(a->u1, a->u2, a->u3, b->u1, b->v1, b->w1);
    c.u2 = dot3(a->u1, a->u2, a->u3, b->u2, b->v2, b->w2);
    c.u3 = dot3(a->u1, a->u2, a->u3, b->u3, b->v3, b->w3);
    c.v1 = dot3(a->v1, a->v2, a->v3, b->u1, b->v1, b->w1);
    c.v2 = dot3(a->v1, a->v2, a->v3, b->u2, b->v2, b->w2);
    c.v3 = dot3(a->v1, a->v2, a->v3, b->u3, b->v3, b->w3);
    c.w1 = dot3(a->w1, a->w2, a->w3, b->u1, b->v1, b->w1);
    c.w2 = dot3(a->w1, a->w2, a->w3, b->u2, b->v2, b->w2);
    c.w3 = dot3(a->w1, a->w2, a->w3, b->u3, b->v3, b->w3);
    return c;
}

static inline qo_cplx conj_(qo_cplx a) { qo_cplx c = { a.re, -a.im }; return c; }

qo_su3 qo_herm3(const qo_su3 *a)
{
    /* :145-170 */
    qo_su3 r;
    r.u1 = conj_(a->u1); r.u2 = conj_(a->v1); r.u3 = conj_(a->w1);
    r.v1 = conj_(a->u2); r.v2 = conj_(a->v2); r.v3 = conj_(a->w2);
    r.w1 = conj_(a->u3); r.w2 = conj_(a->v3); r.w3 = conj_(a->w3);
    return r;
}

static inline void add3(qo_su3 *a, const qo_su3 *b)
{
    /* operator+ :82-98 (a = a + b) */
    qo_cplx *pa = (qo_cplx *)a;
    const qo_cplx *pb = (const qo_cplx *)b;
    for (int i = 0; i < 9; i++) { pa[i].re = pa[i].re + pb[i].re; pa[i].im = pa[i].im + pb[i].im; }
}

static inline double retr3(const qo_su3 *a) { return a->u1.re + a->v2.re + a->w3.re; }

static void unity3(qo_su3 *a)
{
    memset(a, 0, sizeof(*a));
    a->u1.re = 1.0; a->v2.re = 1.0; a->w3.re = 1.0;
}

/* SU(N)/kernel/complex.h:371-376, 346-364 */
static inline qo_cplx cmul(qo_cplx a, qo_cplx b)
{
    qo_cplx c;
    c.re = a.re * b.re - a.im * b.im;
    c.im = a.re * b.im + a.im * b.re;
    return c;
}
static inline qo_cplx cadd(qo_cplx a, qo_cplx b) { qo_cplx c = { a.re + b.re, a.im + b.im }; return c; }
static inline qo_cplx csub(qo_cplx a, qo_cplx b) { qo_cplx c = { a.re - b.re, a.im - b.im }; return c; }

static void reconstruct3(qo_su3 *a)
{
    /* matrix_reconstruct :219-226 -- w = (u x v)* */
    a->w1.re =  a->u3.im * a->v2.im - a->u3.re * a->v2.re - a->u2.im * a->v3.im + a->u2.re * a->v3.re;
    a->w1.im =  a->u3.re * a->v2.im + a->u3.im * a->v2.re - a->u2.re * a->v3.im - a->u2.im * a->v3.re;
    a->w2.re = -a->u3.im * a->v1.im + a->u3.re * a->v1.re + a->u1.im * a->v3.im - a->u1.re * a->v3.re;
    a->w2.im = -a->u3.re * a->v1.im - a->u3.im * a->v1.re + a->u1.re * a->v3.im + a->u1.im * a->v3.re;
    a->w3.re =  a->u2.im * a->v1.im - a->u2.re * a->v1.re - a->u1.im * a->v2.im + a->u1.re * a->v2.re;
    a->w3.im =  a->u2.re * a->v1.im + a->u2.im * a->v1.re - a->u1.re * a->v2.im - a->u1.im * a->v2.re;
}

void qo_gram_schmidt3(qo_su3 *a)
{
    /* GramSchmidt :185-217 */
    qo_su3 r;
    double norm_u = sqrt(a->u1.re * a->u1.re + a->u2.re * a->u2.re + a->u3.re * a->u3.re +
                         a->u1.im * a->u1.im + a->u2.im * a->u2.im + a->u3.im * a->u3.im);
    r.u1.re = a->u1.re / norm_u; r.u1.im = a->u1.im / norm_u;
    r.u2.re = a->u2.re / norm_u; r.u2.im = a->u2.im / norm_u;
    r.u3.re = a->u3.re / norm_u; r.u3.im = a->u3.im / norm_u;

    qo_cplx sp = cadd(cadd(cmul(a->v1, conj_(r.u1)), cmul(a->v2, conj_(r.u2))), cmul(a->v3, conj_(r.u3)));
    r.v1 = csub(a->v1, cmul(r.u1, sp));
    r.v2 = csub(a->v2, cmul(r.u2, sp));
    r.v3 = csub(a->v3, cmul(r.u3, sp));

    double norm_v = sqrt(r.v1.re * r.v1.re + r.v2.re * r.v2.re + r.v3.re * r.v3.re +
                         r.v1.im * r.v1.im + r.v2.im * r.v2.im + r.v3.im * r.v3.im);
    r.v1.re /= norm_v; r.v1.im /= norm_v;
    r.v2.re /= norm_v; r.v2.im /= norm_v;
    r.v3.re /= norm_v; r.v3.im /= norm_v;

    reconstruct3(&r);
    *a = r;
}

void qo_matrix_gid3(qo_su3 *a, int gid, int dir, int sites)
{
    /* lattice_matrixGID :64-80 */
    a->u1.re = sin(((dir + 1) / 1000.0 + 10.0 / sites) * gid) / 2.0;
    a->u1.im = cos(((dir + 1) / 1000.0 - 10.0 / sites) * gid) / 3.0;
    a->u2.re = sin(((dir + 1) / 1000.0 - 18.0 / sites) * gid) / 4.0;
    a->u2.im = cos(((dir + 1) / 1000.0 + 12.0 / sites) * gid) / 4.0;
    a->u3.re = sin(((dir + 1) / 1000.0 - 12.0 / sites) * gid) / 3.0;
    a->u3.im = cos(((dir + 1) / 1000.0 - 15.0 / sites) * gid) / 2.0;

    a->v1.re = sin(((dir + 1) / 1000.0 + 15.0 / sites) * gid) / 3.0;
    a->v1.im = cos(((dir + 1) / 1000.0 - 11.0 / sites) * gid) / 4.0;
    a->v2.re = sin(((dir + 1) / 1000.0 - 11.0 / sites) * gid) / 3.0;
    a->v2.im = cos(((dir + 1) / 1000.0 + 16.0 / sites) * gid) / 2.0;
    a->v3.re = sin(((dir + 1) / 1000.0 - 16.0 / sites) * gid) / 3.0;
    a->v3.im = cos(((dir + 1) / 1000.0 - 18.0 / sites) * gid) / 2.0;

    qo_gram_schmidt3(a);
}

/* ------------------------------------------------------------------------- */
/* SU(2) algebra: SU(N)/suncpp/su2/algebra_su2.cpp                             */

static inline qo_cplx dot2(qo_cplx a1, qo_cplx a2, qo_cplx b1, qo_cplx b2)
{
    /* :93-94 */
    qo_cplx c;
    c.re = -a1.im * b1.im + a1.re * b1.re - a2.im * b2.im + a2.re * b2.re;
    c.im =  a1.re * b1.im + a1.im * b1.re + a2.re * b2.im + a2.im * b2.re;
    return c;
}

qo_su2 qo_mul2(const qo_su2 *a, const qo_su2 *b)
{
    /* :90-102 */
    qo_su2 c;
    c.u1 = dot2(a->u1, a->u2, b->u1, b->v1);
    c.u2 = dot2(a->u1, a->u2, b->u2, b->v2);
    c.v1 = dot2(a->v1, a->v2, b->u1, b->v1);
    c.v2 = dot2(a->v1, a->v2, b->u2, b->v2);
    return c;
}

qo_su2 qo_herm2(const qo_su2 *a)
{
    /* :108-122 */
    qo_su2 r;
    r.u1 = conj_(a->u1); r.u2 = conj_(a->v1);
    r.v1 = conj_(a->u2); r.v2 = conj_(a->v2);
    return r;
}

static inline void add2(qo_su2 *a, const qo_su2 *b)
{
    qo_cplx *pa = (qo_cplx *)a;
    const qo_cplx *pb = (const qo_cplx *)b;
    for (int i = 0; i < 4; i++) { pa[i].re = pa[i].re + pb[i].re; pa[i].im = pa[i].im + pb[i].im; }
}

static inline double retr2(const qo_su2 *a) { return a->u1.re + a->v2.re; }

static void unity2(qo_su2 *a)
{
    memset(a, 0, sizeof(*a));
    a->u1.re = 1.0; a->v2.re = 1.0;
}

void qo_matrix_gid2(qo_su2 *a, int gid, int dir, int sites)
{
    /* :54-64 */
    a->u1.re = sin(((dir + 1) / 1000.0 + 10.0 / sites) * gid) / 2.0;
    a->u2.im = cos(((dir + 1) / 1000.0 - 10.0 / sites) * gid) / 3.0;
    a->u2.re = sin(((dir + 1) / 1000.0 - 18.0 / sites) * gid) / 4.0;
    a->u1.im = sqrt(1 - a->u1.re * a->u1.re - a->u2.im * a->u2.im - a->u2.re * a->u2.re);

    a->v1.re = -a->u2.re;
    a->v1.im =  a->u2.im;
    a->v2.re =  a->u1.re;
    a->v2.im = -a->u1.im;
}

/* ------------------------------------------------------------------------- */
/* heat-bath: SU(N)/suncpp/su2/update_su2.cpp:40-126                            */

static inline void draw(qo_ranlux *rng, float *rnd, int n, unsigned long long *draws)
{
    for (int i = 0; i < n; i++) rnd[i] = qo_ranlux_next(rng);
    if (draws) *draws += (unsigned long long)n;
}

void qo_update_link2(qo_su2 *a, const qo_su2 *stap_in, double beta, int nhit, int gid,
                     int gid_start, int fsites, int final, qo_ranlux *rng,
                     unsigned long long *draws)
{
    qo_su2 stap = *stap_in;   /* by value in the reference: a may alias stap */
    qo_su2 aH, c;
    int flag = 0;
    float rnd[4];

    double Mx, My, Mz, Mw;
    double det, bdet, cosrnd;
    double delta = 0.0;
    double costh, sinth, cosal, sinal, phi, sinphi, cosphi;
    int i = 0;

    Mx = (stap.u1.re + stap.v2.re);
    My = (stap.u1.im - stap.v2.im);
    Mz = (stap.u2.re - stap.v1.re);
    Mw = (stap.u2.im + stap.v1.im);

    det = sqrt(Mx * Mx + My * My + Mz * Mz + Mw * Mw);
    bdet = beta * det;

    aH.u1.re =  Mx / det;
    aH.u1.im = -My / det;
    aH.u2.re = -Mz / det;
    aH.u2.im = -Mw / det;
    aH.v2.re =  aH.u1.re;
    aH.v2.im = -aH.u1.im;
    aH.v1.re = -aH.u2.re;
    aH.v1.im =  aH.u2.im;

    while ((i < nhit) && (flag == 0)) {
        if (gid_start) {
            rnd[0] = fabs((float)sin((0.005 * (1.0 + nhit) + 270.0 / fsites) * gid));
            rnd[1] = fabs((float)cos((0.005 * (1.0 + nhit) +  60.0 / fsites) * gid));
            rnd[2] = fabs((float)sin((0.005 * (1.0 + nhit) - 150.0 / fsites) * gid));
            rnd[3] = fabs((float)cos((0.005 * (1.0 + nhit) - 380.0 / fsites) * gid));
        } else {
            draw(rng, rnd, 4, draws);
        }
        cosrnd = cos(QO_PI2 * rnd[1]);
        if (!gid_start) {
            delta = -(log(1.0 - rnd[0]) + cosrnd * cosrnd * log(1.0 - rnd[2])) / bdet;
            if ((rnd[3] * rnd[3]) <= (1.0 - 0.5 * delta)) flag = 1;
        } else {
            delta = cosrnd * cosrnd / bdet;
            flag = 1;
        }
        i++;
    }

    if (flag) {
        if (gid_start) {
            rnd[0] = fabs((float)cos((0.08 - 270.0 / fsites) * gid));
            rnd[1] = fabs((float)sin((0.08 -  60.0 / fsites) * gid));
        } else {
            draw(rng, rnd, 2, draws);
        }
        cosal = 1.0 - delta;
        costh = 2.0 * rnd[0] - 1.0;
        sinth = sqrt(1.0 - costh * costh);
        sinal = sqrt(1.0 - cosal * cosal);
        phi   = QO_PI2 * rnd[1];

        sinphi = sin(phi);
        cosphi = cos(phi);

        c.u1.re =  cosal;
        c.u1.im =  sinal * costh;
        c.u2.re =  sinal * sinth * sinphi;
        c.u2.im =  sinal * sinth * cosphi;
        c.v2.re =  c.u1.re;
        c.v2.im = -c.u1.im;
        c.v1.re = -c.u2.re;
        c.v1.im =  c.u2.im;

        *a = qo_mul2(&c, &aH);
    }

    if ((!final) && (!flag)) unity2(a);
}

/* Cabibbo-Marinari: SU(N)/suncpp/su3/update_su3.cpp:40-114 */
void qo_update_link3(qo_su3 *U, const qo_su3 *stap, double beta, int nhit, int gid,
                     int gid_start, int fsites, qo_ranlux *rng, unsigned long long *draws)
{
    qo_su3 U0 = *U;
    qo_su3 x0, Vg;
    qo_su2 r0;

    /* rows/cols (1,2) */
    x0 = qo_mul3(&U0, stap);
    r0.u1 = x0.u1; r0.u2 = x0.u2; r0.v1 = x0.v1; r0.v2 = x0.v2;
    qo_update_link2(&r0, &r0, beta, nhit, gid, gid_start, fsites, 0, rng, draws);
    memset(&Vg, 0, sizeof(Vg));
    Vg.u1 = r0.u1; Vg.u2 = r0.u2;
    Vg.v1 = r0.v1; Vg.v2 = r0.v2;
    Vg.w3.re = 1.0;
    U0 = qo_mul3(&Vg, &U0);

    /* rows/cols (1,3) */
    x0 = qo_mul3(&U0, stap);
    r0.u1 = x0.u1; r0.u2 = x0.u3; r0.v1 = x0.w1; r0.v2 = x0.w3;
    qo_update_link2(&r0, &r0, beta, nhit, gid, gid_start, fsites, 0, rng, draws);
    memset(&Vg, 0, sizeof(Vg));
    Vg.u1 = r0.u1; Vg.u3 = r0.u2;
    Vg.v2.re = 1.0;
    Vg.w1 = r0.v1; Vg.w3 = r0.v2;
    U0 = qo_mul3(&Vg, &U0);

    /* rows/cols (2,3) */
    x0 = qo_mul3(&U0, stap);
    r0.u1 = x0.v2; r0.u2 = x0.v3; r0.v1 = x0.w2; r0.v2 = x0.w3;
    qo_update_link2(&r0, &r0, beta, nhit, gid, gid_start, fsites, 0, rng, draws);
    memset(&Vg, 0, sizeof(Vg));
    Vg.u1.re = 1.0;
    Vg.v2 = r0.u1; Vg.v3 = r0.u2;
    Vg.w2 = r0.v1; Vg.w3 = r0.v2;
    U0 = qo_mul3(&Vg, &U0);

    qo_gram_schmidt3(&U0);
    *U = U0;
}

/* ------------------------------------------------------------------------- */
/* Overrelaxation.  ABSENT from the reference ("parity unpinned"): built in the reference's own
 * terms so that it slots into the same pass structure.  In the heat-bath the new SU(2) element
 * is c * aH with aH = (M/det)^+ (update_su2.cpp:56-66) and c drawn at random; the microcanonical
 * move reflects instead of drawing: c = aH, i.e. the element aH*aH.  Re Tr of the 2x2 block of
 * U*A is unchanged by it, so the action is (to rounding) invariant: that is the test. */

static void or_element2(const qo_su2 *stap, qo_su2 *out)
{
    double Mx = (stap->u1.re + stap->v2.re);
    double My = (stap->u1.im - stap->v2.im);
    double Mz = (stap->u2.re - stap->v1.re);
    double Mw = (stap->u2.im + stap->v1.im);
    double det = sqrt(Mx * Mx + My * My + Mz * Mz + Mw * Mw);
    qo_su2 aH;
    aH.u1.re =  Mx / det;
    aH.u1.im = -My / det;
    aH.u2.re = -Mz / det;
    aH.u2.im = -Mw / det;
    aH.v2.re =  aH.u1.re;
    aH.v2.im = -aH.u1.im;
    aH.v1.re = -aH.u2.re;
    aH.v1.im =  aH.u2.im;
    *out = qo_mul2(&aH, &aH);
}

/* SU(2): with W = U*A = c_old * det the reflection is c_new = c_old^+, i.e. U' = (aH_W aH_W) * U
 * with aH_W = (W/det)^+ (equivalently U' = V^+ U^+ V^+, V = A/det) */
void qo_overrelax_link2(qo_su2 *U, const qo_su2 *stap)
{
    qo_su2 W = qo_mul2(U, stap), r;
    or_element2(&W, &r);
    *U = qo_mul2(&r, U);
}

void qo_overrelax_link3(qo_su3 *U, const qo_su3 *stap)
{
    /* the three Cabibbo-Marinari subgroups in the heat-bath's order (update_su3.cpp:45-110) */
    qo_su3 U0 = *U, x0, Vg;
    qo_su2 b, r;

    x0 = qo_mul3(&U0, stap);
    b.u1 = x0.u1; b.u2 = x0.u2; b.v1 = x0.v1; b.v2 = x0.v2;
    or_element2(&b, &r);
    memset(&Vg, 0, sizeof(Vg));
    Vg.u1 = r.u1; Vg.u2 = r.u2; Vg.v1 = r.v1; Vg.v2 = r.v2; Vg.w3.re = 1.0;
    U0 = qo_mul3(&Vg, &U0);

    x0 = qo_mul3(&U0, stap);
    b.u1 = x0.u1; b.u2 = x0.u3; b.v1 = x0.w1; b.v2 = x0.w3;
    or_element2(&b, &r);
    memset(&Vg, 0, sizeof(Vg));
    Vg.u1 = r.u1; Vg.u3 = r.u2; Vg.v2.re = 1.0; Vg.w1 = r.v1; Vg.w3 = r.v2;
    U0 = qo_mul3(&Vg, &U0);

    x0 = qo_mul3(&U0, stap);
    b.u1 = x0.v2; b.u2 = x0.v3; b.v1 = x0.w2; b.v2 = x0.w3;
    or_element2(&b, &r);
    memset(&Vg, 0, sizeof(Vg));
    Vg.u1.re = 1.0; Vg.v2 = r.u1; Vg.v3 = r.u2; Vg.w2 = r.v1; Vg.w3 = r.v2;
    U0 = qo_mul3(&Vg, &U0);

    qo_gram_schmidt3(&U0);
    *U = U0;
}

/* ------------------------------------------------------------------------- */
/* staples and passes: SU(N)/suncpp/Update/sun_update.h                         */

/* ---- twisted boundary condition: the table holds Omega U_y on the slice x = Nx-1 while a pass or a measurement
 * reads it; afterwards every link the pass did not update gets its stored value back */
void qo_set_twist(qo_lattice *lat, double phi, double omega)
{
    lat->phi = phi;
    lat->omega = omega;
}
static int twisted(const qo_lattice *lat) { return lat->phi != 0.0 || lat->omega != 0.0; }
static void twist_begin(const qo_lattice *clat)
{
    qo_lattice *lat = (qo_lattice *)clat;
    if (!twisted(lat) || lat->tw_depth++ > 0) return;
    const int slice = lat->L[1] * lat->L[2] * lat->L[3], first = (lat->L[0] - 1) * slice, nd = lat->nd;
    const size_t esz = (lat->group == 3) ? sizeof(qo_su3) : sizeof(qo_su2);
    lat->tw_backup = malloc((size_t)slice * esz);
    for (int k = 0; k < slice; k++) {
        const int gid = first + k;
        if (lat->group == 3) {
            qo_su3 *m = &((qo_su3 *)lat->links)[gid * nd + 1];
            ((qo_su3 *)lat->tw_backup)[k] = *m;
            const double t1 = 0.5 * (lat->phi + lat->omega), t2 = 0.5 * (lat->omega - lat->phi);
            const qo_cplx p1 = { cos(t1), sin(t1) }, p2 = { cos(t2), sin(t2) };
            m->u1 = cmul(m->u1, p1); m->u2 = cmul(m->u2, p1); m->u3 = cmul(m->u3, p1);
            m->v1 = cmul(m->v1, p2); m->v2 = cmul(m->v2, p2); m->v3 = cmul(m->v3, p2);
            reconstruct3(m);
        } else {
            qo_su2 *m = &((qo_su2 *)lat->links)[gid * nd + 1];
            ((qo_su2 *)lat->tw_backup)[k] = *m;
            const qo_cplx p1 = { cos(0.5 * lat->phi), sin(0.5 * lat->phi) };
            m->u1 = cmul(m->u1, p1); m->u2 = cmul(m->u2, p1);
            m->v1.re = -m->u2.re; m->v1.im = m->u2.im;      /* second row from the first, algebra_su2.cpp:137-142 */
            m->v2.re = m->u1.re;  m->v2.im = -m->u1.im;
        }
    }
}
/* updated_parity >= 0: the pass over direction y on that parity has replaced those links; they stay */
static void twist_end(const qo_lattice *clat, int updated_parity)
{
    qo_lattice *lat = (qo_lattice *)clat;
    if (!twisted(lat) || --lat->tw_depth > 0) return;
    const int slice = lat->L[1] * lat->L[2] * lat->L[3], first = (lat->L[0] - 1) * slice, nd = lat->nd;
    for (int k = 0; k < slice; k++) {
        const int gid = first + k;
        int c[4];
        qo_gid_to_coords(lat->L, (unsigned)gid, c);
        if (updated_parity >= 0 && ((c[0] + c[1] + c[2] + c[3]) & 1) == updated_parity) continue;
        if (lat->group == 3) ((qo_su3 *)lat->links)[gid * nd + 1] = ((qo_su3 *)lat->tw_backup)[k];
        else ((qo_su2 *)lat->links)[gid * nd + 1] = ((qo_su2 *)lat->tw_backup)[k];
    }
    free(lat->tw_backup);
    lat->tw_backup = NULL;
}

void qo_staple3(const qo_lattice *lat, int gid, int dir, qo_su3 *out)
{
    /* :43-77 */
    const qo_su3 *tab = (const qo_su3 *)lat->links;
    const int nd = lat->nd;
    qo_su3 stap, m1, m2, m3, t;
    memset(&stap, 0, sizeof(stap));
    for (int dir1 = 0; dir1 < nd; dir1++) {
        if (dir1 == dir) continue;
        int gid1 = qo_neighbour(lat->L, gid, dir);
        m1 = tab[gid1 * nd + dir1];
        gid1 = qo_neighbour(lat->L, gid, dir1);
        m2 = qo_herm3(&tab[gid1 * nd + dir]);
        m3 = qo_herm3(&tab[gid * nd + dir1]);
        t = qo_mul3(&m1, &m2);
        t = qo_mul3(&t, &m3);
        add3(&stap, &t);

        gid1 = qo_neighbour_back(lat->L, gid, dir1);
        m3 = tab[gid1 * nd + dir1];
        m2 = qo_herm3(&tab[gid1 * nd + dir]);
        gid1 = qo_neighbour(lat->L, gid1, dir);
        m1 = qo_herm3(&tab[gid1 * nd + dir1]);
        t = qo_mul3(&m1, &m2);
        t = qo_mul3(&t, &m3);
        add3(&stap, &t);
    }
    *out = stap;
}

void qo_staple2(const qo_lattice *lat, int gid, int dir, qo_su2 *out)
{
    const qo_su2 *tab = (const qo_su2 *)lat->links;
    const int nd = lat->nd;
    qo_su2 stap, m1, m2, m3, t;
    memset(&stap, 0, sizeof(stap));
    for (int dir1 = 0; dir1 < nd; dir1++) {
        if (dir1 == dir) continue;
        int gid1 = qo_neighbour(lat->L, gid, dir);
        m1 = tab[gid1 * nd + dir1];
        gid1 = qo_neighbour(lat->L, gid, dir1);
        m2 = qo_herm2(&tab[gid1 * nd + dir]);
        m3 = qo_herm2(&tab[gid * nd + dir1]);
        t = qo_mul2(&m1, &m2);
        t = qo_mul2(&t, &m3);
        add2(&stap, &t);

        gid1 = qo_neighbour_back(lat->L, gid, dir1);
        m3 = tab[gid1 * nd + dir1];
        m2 = qo_herm2(&tab[gid1 * nd + dir]);
        gid1 = qo_neighbour(lat->L, gid1, dir);
        m1 = qo_herm2(&tab[gid1 * nd + dir1]);
        t = qo_mul2(&m1, &m2);
        t = qo_mul2(&t, &m3);
        add2(&stap, &t);
    }
    *out = stap;
}

/* stream i seeded like instance i of the reference's device generator: instance 0 takes the CPU generator's seed,
 * instance i the next rand() (SU(N)/random/random.cpp:462-465) */
static void make_streams(qo_lattice *lat)
{
    qo_glibc_rand g;
    qo_srand(&g, lat->randseries);
    const int jseed = qo_rand(&g) % 2147483647;
    const int ns = lat->sites / 2 / lat->share;
    free(lat->streams);
    lat->streams = (qo_ranlux *)calloc((size_t)ns, sizeof(qo_ranlux));
    qo_ranlux_init(&lat->streams[0], jseed, lat->nskip);
    for (int i = 1; i < ns; i++) qo_ranlux_init(&lat->streams[i], qo_rand(&g), lat->nskip);
}
/* the stream of half-site i (= gid / 2: y-pair fastest, then z, t, x) */
static int stream_of_half_site(const qo_lattice *lat, int i)
{
    if (lat->share <= 1) return i;
    const int nyh = lat->L[1] / 2, nz = lat->L[2], nt = lat->L[3], tq = nt / lat->share;
    const int yh = i % nyh, z = (i / nyh) % nz, t = (i / (nyh * nz)) % nt, x = i / (nyh * nz * nt);
    return ((x * tq + t % tq) * nz + z) * nyh + yh;
}

void qo_update_pass(qo_lattice *lat, int parity, int dir)
{
    /* lattice_update_even / lattice_update_odd :79-117 */
    const int nd = lat->nd;
    const int gid_start = (lat->ints == 2);
    const double beta_eff = lat->beta / lat->group;
    twist_begin(lat);
    for (int i = 0; i < lat->sites / 2; i++) {
        int gid = parity ? (int)qo_odd_gid(lat->L, (unsigned)i) : (int)qo_even_gid(lat->L, (unsigned)i);
        qo_ranlux *rng = (lat->rng_mode == QO_RNG_PERSITE) ? &lat->streams[stream_of_half_site(lat, i)] : &lat->seq;
        if (lat->rng_mode == QO_RNG_PERSITE) qo_ranlux_align(rng);
        if (lat->group == 3) {
            qo_su3 *tab = (qo_su3 *)lat->links;
            qo_su3 stap, U;
            qo_staple3(lat, gid, dir, &stap);
            U = tab[gid * nd + dir];
            qo_update_link3(&U, &stap, beta_eff, lat->nhit, gid, gid_start, lat->sites, rng, &lat->rng_draws);
            tab[gid * nd + dir] = U;
        } else {
            qo_su2 *tab = (qo_su2 *)lat->links;
            qo_su2 stap, U;
            qo_staple2(lat, gid, dir, &stap);
            U = tab[gid * nd + dir];
            qo_update_link2(&U, &stap, beta_eff, lat->nhit, gid, gid_start, lat->sites, 1, rng, &lat->rng_draws);
            tab[gid * nd + dir] = U;
        }
    }
    twist_end(lat, dir == 1 ? parity : -1);
}

/* the same pass restricted to the sites with x in [x_begin, x_end): what one slab of a lattice
 * cut along x does (stream i stays tied to the GLOBAL half-site index i) */
void qo_update_pass_range(qo_lattice *lat, int parity, int dir, int x_begin, int x_end)
{
    const int nd = lat->nd;
    const int gid_start = (lat->ints == 2);
    const double beta_eff = lat->beta / lat->group;
    const int slice = lat->L[1] * lat->L[2] * lat->L[3];
    for (int i = x_begin * slice / 2; i < x_end * slice / 2; i++) {
        int gid = parity ? (int)qo_odd_gid(lat->L, (unsigned)i) : (int)qo_even_gid(lat->L, (unsigned)i);
        qo_ranlux *rng = (lat->rng_mode == QO_RNG_PERSITE) ? &lat->streams[stream_of_half_site(lat, i)] : &lat->seq;
        if (lat->rng_mode == QO_RNG_PERSITE) qo_ranlux_align(rng);
        if (lat->group == 3) {
            qo_su3 *tab = (qo_su3 *)lat->links;
            qo_su3 stap, U;
            qo_staple3(lat, gid, dir, &stap);
            U = tab[gid * nd + dir];
            qo_update_link3(&U, &stap, beta_eff, lat->nhit, gid, gid_start, lat->sites, rng, &lat->rng_draws);
            tab[gid * nd + dir] = U;
        } else {
            qo_su2 *tab = (qo_su2 *)lat->links;
            qo_su2 stap, U;
            qo_staple2(lat, gid, dir, &stap);
            U = tab[gid * nd + dir];
            qo_update_link2(&U, &stap, beta_eff, lat->nhit, gid, gid_start, lat->sites, 1, rng, &lat->rng_draws);
            tab[gid * nd + dir] = U;
        }
    }
}

void qo_overrelax_pass(qo_lattice *lat, int parity, int dir)
{
    const int nd = lat->nd;
    for (int i = 0; i < lat->sites / 2; i++) {
        int gid = parity ? (int)qo_odd_gid(lat->L, (unsigned)i) : (int)qo_even_gid(lat->L, (unsigned)i);
        if (lat->group == 3) {
            qo_su3 *tab = (qo_su3 *)lat->links;
            qo_su3 stap;
            qo_staple3(lat, gid, dir, &stap);
            qo_overrelax_link3(&tab[gid * nd + dir], &stap);
        } else {
            qo_su2 *tab = (qo_su2 *)lat->links;
            qo_su2 stap;
            qo_staple2(lat, gid, dir, &stap);
            qo_overrelax_link2(&tab[gid * nd + dir], &stap);
        }
    }
}

void qo_overrelax_sweep(qo_lattice *lat)
{
    for (int dir = 0; dir < 4; dir++) qo_overrelax_pass(lat, 1, dir);
    for (int dir = 0; dir < 4; dir++) qo_overrelax_pass(lat, 0, dir);
}

void qo_sweep(qo_lattice *lat)
{
    /* SU(N)/suncpp/suncpp.h:133-141 */
    for (int dir = 0; dir < 4; dir++) qo_update_pass(lat, 1, dir);
    for (int dir = 0; dir < 4; dir++) qo_update_pass(lat, 0, dir);
}

/* ------------------------------------------------------------------------- */
/* measurements: SU(N)/suncpp/Measurements/Plq.h:46-94, S.h:46-99               */

static void plaquette_sums_range(const qo_lattice *lat, int gid_begin, int gid_end, double *spat, double *temp);

static void plaquette_sums(const qo_lattice *lat, double *spat, double *temp)
{
    plaquette_sums_range(lat, 0, lat->sites, spat, temp);
}

/* raw (spat, temp) sums over the sites with x in [x_begin, x_end) */
void qo_plaquette_sums_range(const qo_lattice *lat, int x_begin, int x_end, double out[2])
{
    const int slice = lat->L[1] * lat->L[2] * lat->L[3];
    plaquette_sums_range(lat, x_begin * slice, x_end * slice, &out[0], &out[1]);
}

static void plaquette_sums_range(const qo_lattice *lat, int gid_begin, int gid_end, double *spat, double *temp)
{
    const int nd = lat->nd;
    double re = 0.0, im = 0.0;
    twist_begin(lat);
    for (int gid = gid_begin; gid < gid_end; gid++) {
        for (int dir = 0; dir < nd - 2; dir++)
            for (int dir1 = dir + 1; dir1 < nd - 1; dir1++) {
                int g1 = qo_neighbour(lat->L, gid, dir);
                int g2 = qo_neighbour(lat->L, gid, dir1);
                if (lat->group == 3) {
                    const qo_su3 *tab = (const qo_su3 *)lat->links;
                    qo_su3 m1 = tab[gid * nd + dir], m2 = tab[g1 * nd + dir1];
                    qo_su3 m3 = qo_herm3(&tab[g2 * nd + dir]), m4 = qo_herm3(&tab[gid * nd + dir1]);
                    qo_su3 t = qo_mul3(&m1, &m2); t = qo_mul3(&t, &m3); t = qo_mul3(&t, &m4);
                    re += retr3(&t);
                } else {
                    const qo_su2 *tab = (const qo_su2 *)lat->links;
                    qo_su2 m1 = tab[gid * nd + dir], m2 = tab[g1 * nd + dir1];
                    qo_su2 m3 = qo_herm2(&tab[g2 * nd + dir]), m4 = qo_herm2(&tab[gid * nd + dir1]);
                    qo_su2 t = qo_mul2(&m1, &m2); t = qo_mul2(&t, &m3); t = qo_mul2(&t, &m4);
                    re += retr2(&t);
                }
            }
        for (int dir = 0; dir < nd - 1; dir++) {
            int g1 = qo_neighbour(lat->L, gid, dir);
            int g2 = qo_neighbour(lat->L, gid, nd - 1);
            if (lat->group == 3) {
                const qo_su3 *tab = (const qo_su3 *)lat->links;
                qo_su3 m1 = tab[gid * nd + dir], m2 = tab[g1 * nd + nd - 1];
                qo_su3 m3 = qo_herm3(&tab[g2 * nd + dir]), m4 = qo_herm3(&tab[gid * nd + nd - 1]);
                qo_su3 t = qo_mul3(&m1, &m2); t = qo_mul3(&t, &m3); t = qo_mul3(&t, &m4);
                im += retr3(&t);
            } else {
                const qo_su2 *tab = (const qo_su2 *)lat->links;
                qo_su2 m1 = tab[gid * nd + dir], m2 = tab[g1 * nd + nd - 1];
                qo_su2 m3 = qo_herm2(&tab[g2 * nd + dir]), m4 = qo_herm2(&tab[gid * nd + nd - 1]);
                qo_su2 t = qo_mul2(&m1, &m2); t = qo_mul2(&t, &m3); t = qo_mul2(&t, &m4);
                im += retr2(&t);
            }
        }
    }
    twist_end(lat, -1);
    *spat = re;
    *temp = im;
}

void qo_plaquette(const qo_lattice *lat, double out[3])
{
    const int nd = lat->nd;
    double re, im;
    plaquette_sums(lat, &re, &im);
    out[2] = (re + im) / ((nd - 1) * nd / 2 * lat->sites);            /* Plq.h:90 */
    out[0] = re / ((nd - 2) * (nd - 1) / 2 * lat->sites);             /* Plq.h:91 */
    out[1] = im / ((nd - 1) * lat->sites);                            /* Plq.h:92 */
}

/* Wilson loop W(R,T): device kernel lattice_measurement_wilson, SU(N)/suncl/wilson_loop.cl:54-453
 * (no CPU twin in the reference: "parity unpinned"; W(1,1) is the time-like plaquette).  Per site p
 * and spatial direction i: ReTr[ Tline(p) Rline_i(p + T t) Tline(p + R i)^+ Rline_i(p)^+ ] with
 * Tline(q) = prod_{k<T} U_t(q + k t), Rline_i(q) = prod_{k<R} U_i(q + k i); average over 3 V
 * (SU(N)/suncl/suncl.cpp:1606-1611, lattice_retrace_plaquette3, su3_measurements_cl.cl:527-547). */
static qo_su3 line3(const qo_lattice *lat, int gid, int dir, int len, int *end)
{
    const qo_su3 *tab = (const qo_su3 *)lat->links;
    qo_su3 m = tab[gid * lat->nd + dir];
    for (int k = 1; k < len; k++) {
        gid = qo_neighbour(lat->L, gid, dir);
        m = qo_mul3(&m, &tab[gid * lat->nd + dir]);
    }
    *end = qo_neighbour(lat->L, gid, dir);
    return m;
}
static qo_su2 line2(const qo_lattice *lat, int gid, int dir, int len, int *end)
{
    const qo_su2 *tab = (const qo_su2 *)lat->links;
    qo_su2 m = tab[gid * lat->nd + dir];
    for (int k = 1; k < len; k++) {
        gid = qo_neighbour(lat->L, gid, dir);
        m = qo_mul2(&m, &tab[gid * lat->nd + dir]);
    }
    *end = qo_neighbour(lat->L, gid, dir);
    return m;
}
double qo_wilson_loop(const qo_lattice *lat, int R, int T)
{
    double sum = 0.0;
    twist_begin(lat);
    for (int gid = 0; gid < lat->sites; gid++) {
        int pT, pR, dummy;
        if (lat->group == 3) {
            qo_su3 m1 = line3(lat, gid, 3, T, &pT);
            for (int i = 0; i < 3; i++) {
                qo_su3 m4 = line3(lat, gid, i, R, &pR);
                qo_su3 m2 = line3(lat, pT, i, R, &dummy);
                qo_su3 m3 = line3(lat, pR, 3, T, &dummy);
                qo_su3 w = qo_mul3(&m1, &m2), h = qo_herm3(&m3);
                w = qo_mul3(&w, &h);
                h = qo_herm3(&m4);
                w = qo_mul3(&w, &h);
                sum += w.u1.re + w.v2.re + w.w3.re;
            }
        } else {
            qo_su2 m1 = line2(lat, gid, 3, T, &pT);
            for (int i = 0; i < 3; i++) {
                qo_su2 m4 = line2(lat, gid, i, R, &pR);
                qo_su2 m2 = line2(lat, pT, i, R, &dummy);
                qo_su2 m3 = line2(lat, pR, 3, T, &dummy);
                qo_su2 w = qo_mul2(&m1, &m2), h = qo_herm2(&m3);
                w = qo_mul2(&w, &h);
                h = qo_herm2(&m4);
                w = qo_mul2(&w, &h);
                sum += w.u1.re + w.v2.re;
            }
        }
    }
    twist_end(lat, -1);
    return sum / (3.0 * lat->sites);
}

void qo_action(const qo_lattice *lat, double out[3])
{
    const int nd = lat->nd;
    const int group = lat->group;
    const double bbeta = lat->beta;
    double re, im;
    plaquette_sums(lat, &re, &im);
    double tot = (re + im) / ((nd - 1) * nd / 2 * lat->sites);        /* S.h:92 */
    tot = bbeta * (1 - tot / group);                                  /* S.h:93 */
    re /= ((nd - 2) * (nd - 1) / 2 * lat->sites);                     /* S.h:94 */
    im /= ((nd - 1) * lat->sites);                                    /* S.h:95 */
    out[0] = bbeta * (1 - re / group);                                /* S.h:96 */
    out[1] = bbeta * (1 - im / group);                                /* S.h:97 */
    out[2] = tot;
}

static void polyakov_sums_range(const qo_lattice *lat, int x_begin, int x_end, double s[4]);

/* raw sums (Re P, Im P, |P|^2, |P|^4) over the spatial sites with x in [x_begin, x_end) */
void qo_polyakov_sums_range(const qo_lattice *lat, int x_begin, int x_end, double out[4])
{
    polyakov_sums_range(lat, x_begin, x_end, out);
}

void qo_polyakov(const qo_lattice *lat, double out[4])
{
    const int *L = lat->L;
    const int g = lat->group;
    double s[4];
    polyakov_sums_range(lat, 0, L[0], s);
    double n123 = (double)L[0] * L[1] * L[2];
    out[0] = s[0] / (n123 * g);
    out[1] = s[1] / (n123 * g);
    out[2] = s[2] / (n123 * g * g);
    out[3] = s[3] / (n123 * g * g * g * g);
}

static void polyakov_sums_range(const qo_lattice *lat, int x_begin, int x_end, double out[4])
{
    /* SU(N)/suncl/su3cpu.cpp:1383-1433 (su2cpu.cpp twin): ordered product of the T links
     * starting at t = 0; normalisation as :1427-1430.  Parity unpinned (see header). */
    const int nd = lat->nd;
    const int *L = lat->L;
    const int g = lat->group;
    double pl = 0.0, pl_im = 0.0, p2 = 0.0, p4 = 0.0;
    for (int x1 = x_begin; x1 < x_end; x1++)
        for (int x2 = 0; x2 < L[1]; x2++)
            for (int x3 = 0; x3 < L[2]; x3++) {
                int c[4] = { x1, x2, x3, 0 };
                double tr_re, tr_im;
                if (g == 3) {
                    const qo_su3 *tab = (const qo_su3 *)lat->links;
                    qo_su3 m2 = tab[qo_coords_to_gid(L, c) * nd + 3];
                    for (int x4 = 1; x4 < L[3]; x4++) {
                        c[3] = x4;
                        const qo_su3 *m1 = &tab[qo_coords_to_gid(L, c) * nd + 3];
                        qo_su3 m3 = qo_mul3(&m2, m1);
                        m2 = m3;
                    }
                    tr_re = m2.u1.re + m2.v2.re + m2.w3.re;
                    tr_im = m2.u1.im + m2.v2.im + m2.w3.im;
                } else {
                    const qo_su2 *tab = (const qo_su2 *)lat->links;
                    qo_su2 m2 = tab[qo_coords_to_gid(L, c) * nd + 3];
                    for (int x4 = 1; x4 < L[3]; x4++) {
                        c[3] = x4;
                        const qo_su2 *m1 = &tab[qo_coords_to_gid(L, c) * nd + 3];
                        qo_su2 m3 = qo_mul2(&m2, m1);
                        m2 = m3;
                    }
                    tr_re = m2.u1.re + m2.v2.re;
                    tr_im = m2.u1.im + m2.v2.im;
                }
                double q2 = tr_re * tr_re + tr_im * tr_im;
                pl += tr_re;
                pl_im += tr_im;
                p2 += q2;
                p4 += q2 * q2;
            }
    out[0] = pl;
    out[1] = pl_im;
    out[2] = p2;
    out[3] = p4;
}

/* Profiles along a space axis (kernels lattice_action_diff_{x,y,z}, SU(N)/suncl/sun_measurements_cl.cl:186-616,
 * and lattice_polyakov_diff_{x,y,z}, SU(N)/suncl/polyakov.cl:142-402): the per-site plaquette sums
 * and Polyakov loops binned by the site's coordinate along `axis` (0 = x, 1 = y, 2 = z).
 * out[2*i], out[2*i+1] = (spat, temp) ReTr sums resp. (Re, Im) loop sums of plane i.  The reference
 * has these on its device path only: "parity unpinned". */
void qo_plaquette_profile_sums(const qo_lattice *lat, int axis, double *out)
{
    for (int i = 0; i < 2 * lat->L[axis]; i++) out[i] = 0.0;
    twist_begin(lat);
    for (int gid = 0; gid < lat->sites; gid++) {
        int c[4];
        double s, t;
        qo_gid_to_coords(lat->L, (unsigned)gid, c);
        plaquette_sums_range(lat, gid, gid + 1, &s, &t);
        out[2 * c[axis]] += s;
        out[2 * c[axis] + 1] += t;
    }
    twist_end(lat, -1);
}

void qo_polyakov_profile_sums(const qo_lattice *lat, int axis, double *out)
{
    const int *L = lat->L;
    const int slice = L[1] * L[2] * L[3];
    for (int i = 0; i < 2 * L[axis]; i++) out[i] = 0.0;
    /* one spatial site at a time through the range routine: a copy of the lattice's x-slice
     * structure is not needed, the routine only reads */
    for (int x1 = 0; x1 < L[0]; x1++)
        for (int x2 = 0; x2 < L[1]; x2++)
            for (int x3 = 0; x3 < L[2]; x3++) {
                int c[4] = { x1, x2, x3, 0 };
                double tr_re = 0.0, tr_im = 0.0;
                if (lat->group == 3) {
                    const qo_su3 *tab = (const qo_su3 *)lat->links;
                    qo_su3 m2 = tab[qo_coords_to_gid(L, c) * lat->nd + 3];
                    for (int x4 = 1; x4 < L[3]; x4++) {
                        c[3] = x4;
                        qo_su3 m3 = qo_mul3(&m2, &tab[qo_coords_to_gid(L, c) * lat->nd + 3]);
                        m2 = m3;
                    }
                    tr_re = m2.u1.re + m2.v2.re + m2.w3.re;
                    tr_im = m2.u1.im + m2.v2.im + m2.w3.im;
                } else {
                    const qo_su2 *tab = (const qo_su2 *)lat->links;
                    qo_su2 m2 = tab[qo_coords_to_gid(L, c) * lat->nd + 3];
                    for (int x4 = 1; x4 < L[3]; x4++) {
                        c[3] = x4;
                        qo_su2 m3 = qo_mul2(&m2, &tab[qo_coords_to_gid(L, c) * lat->nd + 3]);
                        m2 = m3;
                    }
                    tr_re = m2.u1.re + m2.v2.re;
                    tr_im = m2.u1.im + m2.v2.im;
                }
                const int b = axis == 0 ? x1 : (axis == 1 ? x2 : x3);
                out[2 * b] += tr_re;
                out[2 * b + 1] += tr_im;
            }
    (void)slice;
}

/* F_mu_nu traces (SU(N)/suncl/su3cpu.cpp:1212-1381; device: lattice_retrace_plaquette3_F,
 * SU(N)/suncl/su3_measurements_cl.cl:550-589): per site and plane the plaquette matrix
 * P = U_mu(x) U_nu(x+mu) U_mu(x+nu)^+ U_nu(x)^+ times the Gell-Mann matrix lambda_a,
 * F_re = Im Tr(P lambda_a) / 2, F_im = -Re Tr(P lambda_a) / 2, summed over the sites.
 * field 0: planes xy, xz, yz (get_Fmunu); field 1: planes xt, yt, zt (get_F0mu).
 * out layout: for a in (index1, index2), for the three planes: (re, im). */
static qo_su3 gell_mann(int index)
{
    /* su3cpu.cpp:775-920 */
    qo_su3 a;
    memset(&a, 0, sizeof(a));
    switch (index) {
        case 1: a.u2.re = 1.0; a.v1.re = 1.0; break;
        case 2: a.u2.im = -1.0; a.v1.im = 1.0; break;
        case 3: a.u1.re = 1.0; a.v2.re = -1.0; break;
        case 4: a.u3.re = 1.0; a.w1.re = 1.0; break;
        case 5: a.u3.im = -1.0; a.w1.im = 1.0; break;
        case 6: a.v3.re = 1.0; a.w2.re = 1.0; break;
        case 7: a.v3.im = -1.0; a.w2.im = 1.0; break;
        case 8:
            a.u1.re = 0.57735026918962576450914878050196;
            a.v2.re = 0.57735026918962576450914878050196;
            a.w3.re = -1.1547005383792515290182975610039;
            break;
        default: a.u1.re = a.v2.re = a.w3.re = 1.0; break;
    }
    return a;
}

void qo_fmunu_sums(const qo_lattice *lat, int field, int index1, int index2, double out[12])
{
    const int nd = lat->nd;
    const qo_su3 *tab = (const qo_su3 *)lat->links;
    const qo_su3 l1 = gell_mann(index1), l2 = gell_mann(index2);
    static const int planes[2][3][2] = { { {0, 1}, {0, 2}, {1, 2} }, { {0, 3}, {1, 3}, {2, 3} } };
    for (int i = 0; i < 12; i++) out[i] = 0.0;
    twist_begin(lat);
    if (lat->group != 3) {
        /* SU(2): Pauli matrices, SU(N)/suncl/su2cpu.cpp:496-545, 716-870 */
        const qo_su2 *t2 = (const qo_su2 *)lat->links;
        qo_su2 s[2];
        const int idx[2] = { index1, index2 };
        for (int a = 0; a < 2; a++) {
            memset(&s[a], 0, sizeof(qo_su2));
            if (idx[a] == 1) { s[a].u2.re = 1.0; s[a].v1.re = 1.0; }
            else if (idx[a] == 2) { s[a].u2.im = -1.0; s[a].v1.im = 1.0; }
            else { s[a].u1.re = 1.0; s[a].v2.re = -1.0; }
        }
        for (int gid = 0; gid < lat->sites; gid++)
            for (int k = 0; k < 3; k++) {
                const int dir = planes[field][k][0], dir1 = planes[field][k][1];
                int g1 = qo_neighbour(lat->L, gid, dir);
                int g2 = qo_neighbour(lat->L, gid, dir1);
                qo_su2 m1 = t2[gid * nd + dir], m2 = t2[g1 * nd + dir1];
                qo_su2 m3 = qo_herm2(&t2[g2 * nd + dir]), m4 = qo_herm2(&t2[gid * nd + dir1]);
                qo_su2 p = qo_mul2(&m1, &m2); p = qo_mul2(&p, &m3); p = qo_mul2(&p, &m4);
                for (int a = 0; a < 2; a++) {
                    qo_su2 t = qo_mul2(&p, &s[a]);
                    out[6 * a + 2 * k] += (t.u1.im + t.v2.im) / 2;
                    out[6 * a + 2 * k + 1] += -(t.u1.re + t.v2.re) / 2;
                }
            }
        twist_end(lat, -1);
        return;
    }
    for (int gid = 0; gid < lat->sites; gid++)
        for (int k = 0; k < 3; k++) {
            const int dir = planes[field][k][0], dir1 = planes[field][k][1];
            int g1 = qo_neighbour(lat->L, gid, dir);
            int g2 = qo_neighbour(lat->L, gid, dir1);
            qo_su3 m1 = tab[gid * nd + dir], m2 = tab[g1 * nd + dir1];
            qo_su3 m3 = qo_herm3(&tab[g2 * nd + dir]), m4 = qo_herm3(&tab[gid * nd + dir1]);
            qo_su3 p = qo_mul3(&m1, &m2); p = qo_mul3(&p, &m3); p = qo_mul3(&p, &m4);
            qo_su3 t1 = qo_mul3(&p, &l1), t2 = qo_mul3(&p, &l2);
            out[2 * k] += (t1.u1.im + t1.v2.im + t1.w3.im) / 2;
            out[2 * k + 1] += -(t1.u1.re + t1.v2.re + t1.w3.re) / 2;
            out[6 + 2 * k] += (t2.u1.im + t2.v2.im + t2.w3.im) / 2;
            out[6 + 2 * k + 1] += -(t2.u1.re + t2.v2.re + t2.w3.re) / 2;
        }
    twist_end(lat, -1);
}

void qo_estimates(const double *data, int n, double *mean_out, double *var_out)
{
    /* lattice_estimates, analysis_cpp.cpp:81-93 -- sample 0 excluded, biased variance */
    double mean = 0.0, variance = 0.0;
    int length = n - 1;
    for (int i = 1; i < n; i++) mean += data[i] / length;
    for (int i = 1; i < n; i++) variance += pow((data[i] - mean), 2) / length;
    *mean_out = mean;
    *var_out = variance;
}

/* ------------------------------------------------------------------------- */
/* life cycle                                                                   */

qo_lattice *qo_create(int group, const int L[4], double beta, int nhit, int ints,
                      int rng_mode, int lux_level, unsigned randseries)
{
    qo_lattice *lat = (qo_lattice *)calloc(1, sizeof(qo_lattice));
    lat->group = group;
    lat->nd = 4;
    for (int i = 0; i < 4; i++) lat->L[i] = L[i];
    lat->sites = L[0] * L[1] * L[2] * L[3];
    lat->beta = beta;
    lat->nhit = nhit;
    lat->ints = ints;
    lat->rng_mode = rng_mode;
    lat->nskip = qo_nskip_for_level(lux_level);
    lat->randseries = randseries;
    size_t esz = (group == 3) ? sizeof(qo_su3) : sizeof(qo_su2);
    lat->links = calloc((size_t)lat->nd * lat->sites, esz);   /* sunh.h:101-104 */

    /* PRNG::initialize_CPU, random.cpp:174-198 */
    qo_glibc_rand g;
    qo_srand(&g, randseries);
    int jseed = qo_rand(&g) % 2147483647;
    qo_ranlux_init(&lat->seq, jseed, lat->nskip);
    lat->share = 1;
    if (rng_mode == QO_RNG_PERSITE) make_streams(lat);
    return lat;
}

/* Per-thread streams shared by `share` sites of one parity: the half-sites (x, t' + k Nt/share, z, y-pair),
 * k = 0 .. share-1, use stream ((x Nt/share + t') Nz + z) Ny/2 + y-pair one after the other in every pass and in
 * the hot start (a device lane that updates them in turn moves the generator state once).  Call before the first
 * random number is drawn (before qo_init_links for a hot start).  share must divide Nt. */
int qo_set_rng_share(qo_lattice *lat, int share)
{
    if (share < 1 || lat->L[3] % share) return -1;
    lat->share = share;
    if (lat->rng_mode == QO_RNG_PERSITE) make_streams(lat);
    return 0;
}

void qo_destroy(qo_lattice *lat)
{
    if (!lat) return;
    free(lat->links);
    free(lat->streams);
    free(lat);
}

/* Hot start, device path only in the reference (the CPU build just prints, sunh.h:114-116).
 * SU(3): product of three embedded random SU(2) elements, lattice_random3,
 * SU(N)/suncl/su3_update_cl.cl:164-225 (r[0..3] = rnd1, r[4..7] = rnd2, r[8..11] = rnd3; the .w
 * lanes are not used); SU(2): lattice_random2, SU(N)/suncl/su2_update_cl.cl:109-146.  Constants
 * PI, PI2: SU(N)/suncl/model.cl:120-123.  "parity unpinned": no runnable reference for it here. */
#define QO_PI  3.1415926535897932384626433832795
#define QO_PI2 6.2831853071795864769252867665590
void qo_hot_link3(qo_su3 *m, const float r[12])
{
    double a0[3], a1[3], a2[3], a3[3];
    for (int k = 0; k < 3; k++) {
        const double alpha = QO_PI * (double)r[k];
        const double phi = QO_PI2 * (double)r[4 + k];
        const double costh = 2.0 * (double)r[8 + k] - 1.0;
        const double sinal = sin(alpha);
        a0[k] = cos(alpha);
        const double sinph = sin(phi), cosph = cos(phi);
        const double sinth = sqrt(1.0 - costh * costh);
        const double t1 = sinal * sinth;
        a1[k] = t1 * cosph;
        a2[k] = t1 * sinph;
        a3[k] = sinal * costh;
    }
    enum { x = 0, y = 1, z = 2 };
    m->u1.re = a0[x] * a0[y] - a3[x] * a3[y];
    m->u1.im = a0[y] * a3[x] + a0[x] * a3[y];
    m->u2.re = -a0[x] * a1[y] * a1[z] + a0[z] * a2[x] - a0[x] * a2[y] * a2[z] - a1[z] * a2[y] * a3[x] + a1[y] * a2[z] * a3[x] - a1[x] * a3[z];
    m->u2.im = a0[z] * a1[x] + a0[x] * a1[z] * a2[y] - a0[x] * a1[y] * a2[z] - a1[y] * a1[z] * a3[x] - a2[y] * a2[z] * a3[x] + a2[x] * a3[z];
    m->u3.re = -a1[x] * a1[z] + a0[x] * a0[z] * a2[y] + a2[x] * a2[z] - a0[z] * a1[y] * a3[x] + a0[x] * a1[y] * a3[z] + a2[y] * a3[x] * a3[z];
    m->u3.im = a0[x] * a0[z] * a1[y] + a1[z] * a2[x] + a1[x] * a2[z] + a0[z] * a2[y] * a3[x] - a0[x] * a2[y] * a3[z] + a1[y] * a3[x] * a3[z];
    m->v3.re = -a0[z] * a1[x] * a1[y] - a0[z] * a2[x] * a2[y] + a0[x] * a2[z] + a1[z] * a3[x] - a1[y] * a2[x] * a3[z] + a1[x] * a2[y] * a3[z];
    m->v3.im = a0[x] * a1[z] - a0[z] * a1[y] * a2[x] + a0[z] * a1[x] * a2[y] - a2[z] * a3[x] + a1[x] * a1[y] * a3[z] + a2[x] * a2[y] * a3[z];
    m->v1.re = -a0[y] * a2[x] - a1[x] * a3[y];
    m->v1.im = a0[y] * a1[x] - a2[x] * a3[y];
    m->v2.re = a0[x] * a0[z] + a1[y] * a1[z] * a2[x] - a1[x] * a1[z] * a2[y] + a1[x] * a1[y] * a2[z] + a2[x] * a2[y] * a2[z] + a3[x] * a3[z];
    m->v2.im = -a1[x] * a1[y] * a1[z] - a1[z] * a2[x] * a2[y] + a1[y] * a2[x] * a2[z] - a1[x] * a2[y] * a2[z] - a0[z] * a3[x] + a0[x] * a3[z];
    reconstruct3(m);
}
void qo_hot_link2(qo_su2 *m, const float r[4])
{
    const double alpha = QO_PI * (double)r[0];
    const double phi = QO_PI2 * (double)r[1];
    const double costh = 2.0 * (double)r[2] - 1.0;
    const double sinal = sin(alpha);
    const double a0 = cos(alpha);
    const double sinph = sin(phi), cosph = cos(phi);
    const double sinth = sqrt(1.0 - costh * costh);
    const double t1 = sinal * sinth;
    const double a1 = t1 * cosph, a2 = t1 * sinph, a3 = sinal * costh;
    /* uv1 = (a0, a2, a3, a1) = (Re u1, Re u2, Im u1, Im u2) */
    m->u1.re = a0; m->u1.im = a3;
    m->u2.re = a2; m->u2.im = a1;
    /* second row, lattice_reconstruct2, su3_update_cl.cl:112-128 */
    m->v1.re = -m->u2.re; m->v1.im = m->u2.im;
    m->v2.re = m->u1.re;  m->v2.im = -m->u1.im;
}

/* Hot start with per-thread streams: for each direction X, Y, Z, T in turn, stream h opens a
 * fresh batch of 24; its first 12 values (SU(2): 4) make the link of site 2h, the next 12
 * (SU(2): 4) that of site 2h+1.  The reference's device path likewise produces one fresh buffer
 * per direction (SU(N)/suncl/suncl.cpp:5236-5250) and reads float4 k of site gid at
 * prns[gid + k*ROWSIZE] (SU(N)/suncl/suncl.cl:55-161). */
static void init_hot_persite(qo_lattice *lat)
{
    const int nd = lat->nd, per_link = (lat->group == 3) ? 12 : 4;
    for (int dir = 0; dir < nd; dir++)
        for (int h = 0; h < lat->sites / 2; h++) {
            qo_ranlux *rng = &lat->streams[stream_of_half_site(lat, h)];
            qo_ranlux_align(rng);
            for (int s = 0; s < 2; s++) {
                float r[12];
                for (int k = 0; k < per_link; k++) r[k] = qo_ranlux_next(rng);
                lat->rng_draws += (unsigned long long)per_link;
                const int gid = 2 * h + s;
                if (lat->group == 3) qo_hot_link3(&((qo_su3 *)lat->links)[gid * nd + dir], r);
                else qo_hot_link2(&((qo_su2 *)lat->links)[gid * nd + dir], r);
            }
        }
}

void qo_init_links(qo_lattice *lat)
{
    /* lattice_initializeCPU, sunh.h:111-131.  ints == 0 (hot) only prints in the reference
     * and leaves the calloc'ed zeros in place. */
    const int nd = lat->nd;
    if (lat->ints == 1) {
        for (int i = 0; i < nd * lat->sites; i++) {
            if (lat->group == 3) unity3(&((qo_su3 *)lat->links)[i]);
            else unity2(&((qo_su2 *)lat->links)[i]);
        }
    } else if (lat->ints == 0 && lat->rng_mode == QO_RNG_PERSITE) {
        init_hot_persite(lat);
    } else if (lat->ints == 2) {
        for (int gid = 0; gid < lat->sites; gid++)
            for (int dir = 0; dir < nd; dir++) {
                if (lat->group == 3) qo_matrix_gid3(&((qo_su3 *)lat->links)[gid * nd + dir], gid, dir, lat->sites);
                else qo_matrix_gid2(&((qo_su2 *)lat->links)[gid * nd + dir], gid, dir, lat->sites);
            }
    }
}

void qo_simulate(qo_lattice *lat, int nav, int iter, int niter, double *plq, double *act)
{
    /* lattice_simulateCPU, suncpp.h:124-165 */
    qo_init_links(lat);
    if (plq) qo_plaquette(lat, &plq[0]);
    if (act) qo_action(lat, &act[0]);
    for (int n = 0; n < nav; n++) qo_sweep(lat);
    for (int i = 1; i < iter; i++) {
        for (int n = 0; n < niter; n++) qo_sweep(lat);
        if (plq) qo_plaquette(lat, &plq[3 * i]);
        if (act) qo_action(lat, &act[3 * i]);
    }
}

/* ------------------------------------------------------------------------- */
/* reference device layout                                                      */

void qo_export_device_layout(const qo_lattice *lat, double *table, int rowsize)
{
    const int nd = lat->nd;
    for (int gid = 0; gid < lat->sites; gid++)
        for (int dir = 0; dir < nd; dir++) {
            if (lat->group == 3) {
                const qo_su3 *m = &((const qo_su3 *)lat->links)[gid * nd + dir];
                double *q0 = &table[((size_t)(0 * 4 + dir) * rowsize + gid) * 4];
                double *q1 = &table[((size_t)(1 * 4 + dir) * rowsize + gid) * 4];
                double *q2 = &table[((size_t)(2 * 4 + dir) * rowsize + gid) * 4];
                q0[0] = m->u1.re; q0[1] = m->u2.re; q0[2] = m->u3.re; q0[3] = m->v3.re;
                q1[0] = m->u1.im; q1[1] = m->u2.im; q1[2] = m->u3.im; q1[3] = m->v3.im;
                q2[0] = m->v1.re; q2[1] = m->v2.re; q2[2] = m->v1.im; q2[3] = m->v2.im;
            } else {
                const qo_su2 *m = &((const qo_su2 *)lat->links)[gid * nd + dir];
                double *q0 = &table[((size_t)dir * rowsize + gid) * 4];
                q0[0] = m->u1.re; q0[1] = m->u2.re; q0[2] = m->u1.im; q0[3] = m->u2.im;
            }
        }
}

void qo_import_device_layout(qo_lattice *lat, const double *table, int rowsize)
{
    const int nd = lat->nd;
    for (int gid = 0; gid < lat->sites; gid++)
        for (int dir = 0; dir < nd; dir++) {
            if (lat->group == 3) {
                qo_su3 *m = &((qo_su3 *)lat->links)[gid * nd + dir];
                const double *q0 = &table[((size_t)(0 * 4 + dir) * rowsize + gid) * 4];
                const double *q1 = &table[((size_t)(1 * 4 + dir) * rowsize + gid) * 4];
                const double *q2 = &table[((size_t)(2 * 4 + dir) * rowsize + gid) * 4];
                m->u1.re = q0[0]; m->u2.re = q0[1]; m->u3.re = q0[2]; m->v3.re = q0[3];
                m->u1.im = q1[0]; m->u2.im = q1[1]; m->u3.im = q1[2]; m->v3.im = q1[3];
                m->v1.re = q2[0]; m->v2.re = q2[1]; m->v1.im = q2[2]; m->v2.im = q2[3];
                reconstruct3(m);
            } else {
                qo_su2 *m = &((qo_su2 *)lat->links)[gid * nd + dir];
                const double *q0 = &table[((size_t)dir * rowsize + gid) * 4];
                m->u1.re = q0[0]; m->u2.re = q0[1]; m->u1.im = q0[2]; m->u2.im = q0[3];
                m->v1.re = -m->u2.re; m->v1.im = m->u2.im;
                m->v2.re =  m->u1.re; m->v2.im = -m->u1.im;
            }
        }
}
