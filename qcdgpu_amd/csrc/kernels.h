// kernels.h -- HIP kernels of the SU(N) sweep engine (gfx950).
//
//   k_update      one (parity, direction) heat-bath pass: staple gather + Cabibbo-Marinari /
//                 SU(2) heat-bath + re-unitarisation + in-register RANLUX, one thread per link
//   k_plaquette   6 plaquettes per site, fp64, block partials; k_reduce sums them in a fixed order
//   k_polyakov    ordered product of the Nt time links per spatial site
//   k_reunit      stand-alone re-unitarisation pass
//   k_import/k_export  reference device layout <-> this engine's layout
//   k_halo        boundary slice <-> contiguous message buffer
//
// All of them are bandwidth/latency kernels on byte-addressed HBM data: 16 B per lane per load,
// consecutive lanes on consecutive half-sites.
#pragma once

#include <hip/hip_runtime.h>

#include "lattice.h"
#include "ranlux.h"
#include "su_math.h"

namespace qcd {

// delivered RANLUX batches held in LDS per thread by the update kernel
constexpr int RLX_WINDOW = 2;

// ---------------------------------------------------------------------------------------------
// random sources of the heat-bath
//
// Consumption rule of a per-thread stream: every link update starts at the first value of a
// fresh batch of 24; what the previous update left of its last batch is dropped.  (The
// reference's device path has the same rule in another form: each update kernel reads from the
// front of a buffer produced just for it, SU(N)/suncl/su3_update_cl.cl:679 and
// SU(N)/suncl/suncl.cpp:5357-5377.)  The persistent state of a stream is therefore just the
// generator state in front of its next batch.

struct RlxView {
    unsigned* slot0;    // [w * ns + stream], w in [0, 25): generator state in front of a batch
    unsigned* slot1;    // second copy; ctl says which of the two is current
    unsigned* ctl;      // bit 0: slot that holds the state in front of the stream's next batch
    int ns;             // streams
    int q, rem;         // nskip = 24*q + rem
};

// Where a stream's persistent state lives, and the moves both stream types share.
struct RlxHome {
    unsigned* __restrict__ s0;   // this stream's column in slot 0 / slot 1
    unsigned* __restrict__ s1;
    unsigned* __restrict__ ctl;
    int ns, q, rem;

    __device__ __forceinline__ void bind(const RlxView& view, int stream)
    {
        s0 = view.slot0 + stream;
        s1 = view.slot1 + stream;
        ctl = view.ctl + stream;
        ns = view.ns;
        q = view.q;
        rem = view.rem;
    }
    __device__ __forceinline__ void load_state(RlxRegs& r, int slot) const
    {
        const unsigned* __restrict__ p = slot ? s1 : s0;
#pragma unroll
        for (int w = 0; w < 24; w++) r.s[w] = p[(size_t)w * ns];
        r.c = p[(size_t)24 * ns];
    }
    __device__ __forceinline__ void store_state(const RlxRegs& r, int slot) const
    {
        unsigned* __restrict__ p = slot ? s1 : s0;
#pragma unroll
        for (int w = 0; w < 24; w++) p[(size_t)w * ns] = r.s[w];
        p[(size_t)24 * ns] = r.c;
    }
};

// The stream as the update kernel normally uses it: two delivered batches (48 values) are
// written to LDS column `tid` when the stream is opened, and the generator is dead after that --
// no registers held, no branch into generator code while the heat-bath runs.  An SU(3) link
// update needs 18 values plus 4 per rejected trial, so the window covers seven rejections.  A
// thread that runs dry reports it (close() < 0) and the kernel repeats that one link with the
// general stream below, starting from the same persistent state.
template <int BLOCK, int NSKIP>
struct RlxFast {
    __device__ __forceinline__ void skip(RlxRegs& r) const
    {
        if (NSKIP > 0) rlx_skip_static<(NSKIP > 0 ? NSKIP : 24)>(r);
        else rlx_skip(r, h.q, h.rem);
    }
    const float* buf;
    RlxHome h;
    int pos, cur;
    bool dry;

    __device__ __forceinline__ void open(const RlxView& view, int stream, float* lds_column)
    {
        h.bind(view, stream);
        buf = lds_column;
        dry = false;
        pos = 0;
        cur = (int)(*h.ctl & 1u);
#ifdef QCD_ABL_RNG  // diagnostic build: no generator work, constant numbers
        return;
#endif
        RlxRegs r;
        h.load_state(r, cur);
        rlx_batch(r, lds_column, BLOCK);
        skip(r);
        h.store_state(r, cur ^ 1);   // in front of the second batch: the usual next state
        rlx_batch(r, lds_column + 24 * BLOCK, BLOCK);
    }
    __device__ __forceinline__ float next()
    {
        if (pos >= 24 * RLX_WINDOW) {
            dry = true;
            return 0.5f;
        }
#ifdef QCD_ABL_RNG
        return 0.37f + 0.001f * (float)(pos++ % 24);
#endif
        return buf[(pos++) * BLOCK];
    }
    __device__ __forceinline__ void draw4(float r[4])
    {
        r[0] = next(); r[1] = next(); r[2] = next(); r[3] = next();
    }
    __device__ __forceinline__ void draw2(float r[2])
    {
        r[0] = next(); r[1] = next();
    }
    // number of values used, or -1 when the window ran dry (nothing was made persistent then)
    __device__ __forceinline__ int close()
    {
        if (dry) return -1;
        if (pos > 24) {
            // rare (two or more rejected trials): the second batch was touched, the stream
            // resumes in front of the third; rebuild that state from the one in front of the second
            RlxRegs r;
            h.load_state(r, cur ^ 1);
            rlx_round(r);
            skip(r);
            h.store_state(r, cur);
        } else {
            *h.ctl = (unsigned)(cur ^ 1);
        }
        return pos;
    }
};

// The general stream: any number of values, two delivered batches in LDS used as a ring and
// refilled on demand.  Only threads whose fast window ran dry come here (many rejected trials:
// small beta), so this path is written for size, not speed.
template <int BLOCK>
struct RlxStream {
    float* buf;
    RlxHome h;
    int pos, nb, cur;   // nb batches delivered so far; state in front of batch k is in slot cur^(k&1)
                        // for k = nb-1 and k = nb-2

    __device__ __forceinline__ void open(const RlxView& view, int stream, float* lds_column)
    {
        h.bind(view, stream);
        buf = lds_column;
        pos = 0;
        cur = (int)(*h.ctl & 1u);
#ifdef QCD_ABL_RNG  // diagnostic build: no generator work, constant numbers
        nb = 1000;
        return;
#endif
        RlxRegs r;
        h.load_state(r, cur);
        rlx_batch(r, buf, BLOCK);
        rlx_skip(r, h.q, h.rem);
        h.store_state(r, cur ^ 1);
        rlx_batch(r, buf + 24 * BLOCK, BLOCK);
        nb = 2;
    }
    // state in front of batch nb from the one in front of batch nb-1
    __device__ __forceinline__ void advance(RlxRegs& r)
    {
        h.load_state(r, cur ^ ((nb - 1) & 1));
        rlx_round(r);
        rlx_skip(r, h.q, h.rem);
        h.store_state(r, cur ^ (nb & 1));
    }
    // both delivered batches are used up: deliver the next one over the older
    __device__ __noinline__ void refill()
    {
        RlxRegs r;
        advance(r);
        rlx_batch(r, buf + (nb & 1) * 24 * BLOCK, BLOCK);
        nb++;
    }
    __device__ __forceinline__ float next()
    {
        if (pos == 24 * nb) refill();
        const int b = pos / 24;
        const int k = pos - 24 * b;
        pos++;
#ifdef QCD_ABL_RNG
        return 0.37f + 0.001f * (float)k;
#endif
        return buf[((b & 1) * 24 + k) * BLOCK];
    }
    __device__ __forceinline__ void draw4(float r[4])
    {
        r[0] = next(); r[1] = next(); r[2] = next(); r[3] = next();
    }
    __device__ __forceinline__ void draw2(float r[2])
    {
        r[0] = next(); r[1] = next();
    }
    __device__ __forceinline__ int close()
    {
#ifndef QCD_ABL_RNG
        // the stream resumes in front of the batch after the last one touched
        const int m = (pos - 1) / 24;
        if (m == nb - 1) {
            RlxRegs r;
            advance(r);
            *h.ctl = (unsigned)(cur ^ (nb & 1));
        } else {
            *h.ctl = (unsigned)(cur ^ ((m + 1) & 1));
        }
#endif
        return pos;
    }
};

// INTS = 2: every "random" number is a closed form of the site index
// (SU(N)/suncpp/su2/update_su2.cpp:71-76, 95-98); the values are tabulated on the host
struct GidSource {
    float trial[4];
    float accept[2];
    __device__ __forceinline__ void draw4(float r[4]) const
    {
        r[0] = trial[0]; r[1] = trial[1]; r[2] = trial[2]; r[3] = trial[3];
    }
    __device__ __forceinline__ void draw2(float r[2]) const
    {
        r[0] = accept[0]; r[1] = accept[1];
    }
};

// ---------------------------------------------------------------------------------------------
// SU(2) heat-bath on a 2x2 block (SU(N)/suncpp/su2/update_su2.cpp:40-126).
// Returns false when no trial was accepted within nhit (the caller decides what that means:
// identity inside SU(3), link unchanged for SU(2)).
//
// The accept test of a trial is  r3^2 <= 1 - delta/2,  delta = -(ln(1-r0) + cos^2(2 pi r1) ln(1-r2)) / bdet.
// Lanes of a wavefront need different numbers of trials, so the trial loop runs for the slowest
// lane; to keep that loop cheap the test is first made with the fp32 hardware transcendentals
// and decided there unless it falls within a margin that covers their error (then, and only
// then, the fp64 expression of the reference is evaluated inside the loop).  The fp64 delta of
// the ACCEPTED trial is computed once after the loop.  Decisions are therefore exactly those of
// the fp64 test.  Error bounds behind the margin (measured over all 2^24 arguments on gfx950 by
// tools/fp32_bounds.hip: cos 1.3e-7 absolute, log 1.7e-7 relative) enter with a factor >= 4.
#define QCD_SCREEN_LOG_REL 1.0e-6f
#define QCD_SCREEN_COS_ABS 1.0e-6f

// cos(2 pi r), sin(2 pi r) for r in [0, 1): the reference forms 2*pi*r in fp64 and calls cos/sin;
// cospi/sincospi reduce the argument exactly and are half the instructions
__device__ __forceinline__ double cos_2pi(float r) { return cospi(2.0 * (double)r); }

template <bool GID, class Rng>
__device__ __forceinline__ bool heatbath_su2(const su2& stap, double beta, int nhit, Rng& rng, su2& out)
{
    const double Mx = (stap.e[0][0].re + stap.e[1][1].re);
    const double My = (stap.e[0][0].im - stap.e[1][1].im);
    const double Mz = (stap.e[0][1].re - stap.e[1][0].re);
    const double Mw = (stap.e[0][1].im + stap.e[1][0].im);

    const double det = sqrt(Mx * Mx + My * My + Mz * Mz + Mw * Mw);
    const double bdet = beta * det;
    const double idet = 1.0 / det;

    su2 aH;
    aH.e[0][0] = mk(Mx * idet, -My * idet);
    aH.e[0][1] = mk(-Mz * idet, -Mw * idet);
    aH.e[1][1] = mk(aH.e[0][0].re, -aH.e[0][0].im);
    aH.e[1][0] = mk(-aH.e[0][1].re, aH.e[0][1].im);

    bool flag = false;
    double delta = 0.0;
    float rnd[4];
    if (GID) {
        // INTS = 2: the first trial is always accepted (update_su2.cpp:87-90)
        rng.draw4(rnd);
        const double cosrnd = cos_2pi(rnd[1]);
        delta = cosrnd * cosrnd / bdet;
        flag = true;
    } else {
        const float ibdet32 = (float)(1.0 / bdet);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;  // r0, r1, r2 of the accepted trial
        int i = 0;
        while ((i < nhit) && !flag) {
            rng.draw4(rnd);
            const float c32 = __builtin_amdgcn_cosf(rnd[1]);
            const float l0 = __builtin_amdgcn_logf(1.0f - rnd[0]) * 0.6931471805599453f;
            const float l2 = __builtin_amdgcn_logf(1.0f - rnd[2]) * 0.6931471805599453f;
            const float d32 = -(l0 + c32 * c32 * l2) * ibdet32;
            const float r3sq = rnd[3] * rnd[3];
            const float t32 = 1.0f - 0.5f * d32 - r3sq;
            const float margin = ibdet32 * (4.0f * QCD_SCREEN_LOG_REL * (fabsf(l0) + fabsf(l2)) +
                                            4.0f * QCD_SCREEN_COS_ABS * fabsf(l2) + 1.0e-6f) +
                                 4.0e-6f * fabsf(d32) + 4.0e-6f;
            bool acc = t32 > margin;
            if (fabsf(t32) <= margin) {
                const double cosrnd = cos_2pi(rnd[1]);
                const double d = -(log(1.0 - rnd[0]) + cosrnd * cosrnd * log(1.0 - rnd[2])) / bdet;
                acc = (r3sq <= (1.0 - 0.5 * d));
            }
            if (acc) {
                flag = true;
                a0 = rnd[0];
                a1 = rnd[1];
                a2 = rnd[2];
            }
            i++;
        }
        if (flag) {
            const double cosrnd = cos_2pi(a1);
            delta = -(log(1.0 - a0) + cosrnd * cosrnd * log(1.0 - a2)) / bdet;
        }
    }
    if (flag) {
        rng.draw2(rnd);
        const double cosal = 1.0 - delta;
        const double costh = 2.0 * rnd[0] - 1.0;
        const double sinth = sqrt(1.0 - costh * costh);
        const double sinal = sqrt(1.0 - cosal * cosal);
        double sinphi, cosphi;
        sincospi(2.0 * (double)rnd[1], &sinphi, &cosphi);
        su2 c;
        c.e[0][0] = mk(cosal, sinal * costh);
        c.e[0][1] = mk(sinal * sinth * sinphi, sinal * sinth * cosphi);
        c.e[1][1] = mk(c.e[0][0].re, -c.e[0][0].im);
        c.e[1][0] = mk(-c.e[0][1].re, c.e[0][1].im);
        out = mul(c, aH);
    }
    return flag;
}

// one Cabibbo-Marinari step on rows/columns (P, Q) (SU(N)/suncpp/su3/update_su3.cpp:45-66):
// only the 2x2 block of U*A that the subgroup needs is formed, and only rows P and Q of U
// change under the embedded SU(2) element (the other products of the reference are with exact
// zeros and ones).
template <int P, int Q, bool GID, class Rng>
__device__ __forceinline__ void cm_step(su3& U, const su3& A, double beta, int nhit, Rng& rng)
{
    su2 blk;
    blk.e[0][0] = dot3(U.e[P][0], U.e[P][1], U.e[P][2], A.e[0][P], A.e[1][P], A.e[2][P]);
    blk.e[0][1] = dot3(U.e[P][0], U.e[P][1], U.e[P][2], A.e[0][Q], A.e[1][Q], A.e[2][Q]);
    blk.e[1][0] = dot3(U.e[Q][0], U.e[Q][1], U.e[Q][2], A.e[0][P], A.e[1][P], A.e[2][P]);
    blk.e[1][1] = dot3(U.e[Q][0], U.e[Q][1], U.e[Q][2], A.e[0][Q], A.e[1][Q], A.e[2][Q]);
    su2 r;
    if (heatbath_su2<GID>(blk, beta, nhit, rng, r)) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const cplx up = U.e[P][j], uq = U.e[Q][j];
            U.e[P][j] = dot2(r.e[0][0], r.e[0][1], up, uq);
            U.e[Q][j] = dot2(r.e[1][0], r.e[1][1], up, uq);
        }
    }
}

template <bool GID, class Rng>
__device__ __forceinline__ void update_link(su3& U, const su3& A, double beta, int nhit, Rng& rng)
{
#ifdef QCD_ABL_HB  // diagnostic build: no heat-bath, keeps loads, staple products and the store
    add_to(U, A);
    gram_schmidt(U);
    return;
#endif
    cm_step<0, 1, GID>(U, A, beta, nhit, rng);
    cm_step<0, 2, GID>(U, A, beta, nhit, rng);
    cm_step<1, 2, GID>(U, A, beta, nhit, rng);
    gram_schmidt(U);
}
// SU(2): the new link does not depend on the old one; unchanged when no trial is accepted
// (update_su2.cpp:121-125 with final = 1, SU(N)/suncpp/Update/sun_update.h:94)
template <bool GID, class Rng>
__device__ __forceinline__ void update_link(su2& U, const su2& A, double beta, int nhit, Rng& rng)
{
    su2 r;
    if (heatbath_su2<GID>(A, beta, nhit, rng, r)) U = r;
}

// ---------------------------------------------------------------------------------------------
// staple sum (SU(N)/suncpp/Update/sun_update.h:43-77): nu ascending, forward then backward

// hb = 16 * half index of x, par = parity of x (wave-uniform)
template <int GROUP, int DIR>
__device__ __forceinline__ mat<GROUP> staple(const double2* __restrict__ U, const Geom& g, const int hb,
                                             const NbrOff& n, const int par)
{
    using M = mat<GROUP>;
    M stap = zero<GROUP>();
    const int fmu = off_fwd(n, DIR);
    // not unrolled on purpose: one nu at a time keeps six links (not eighteen) in flight per
    // lane, which is what lets two waves per SIMD fit in the register file
#pragma unroll 1
    for (int k = 0; k < 3; k++) {
        const int nu = (k < DIR) ? k : k + 1;
        const int fnu = off_fwd(n, nu), bnu = off_bwd(n, nu);
        const __amdgpu_buffer_rsrc_t r_nu_o = link_rsrc<GROUP>(U, g, nu, par ^ 1);   // U_nu, other parity
        const __amdgpu_buffer_rsrc_t r_nu_s = link_rsrc<GROUP>(U, g, nu, par);       // U_nu, same parity
        const __amdgpu_buffer_rsrc_t r_mu_o = link_rsrc<GROUP>(U, g, DIR, par ^ 1);  // U_mu, other parity
        {
            // U_nu(x+mu) U_mu(x+nu)^+ U_nu(x)^+
            const M m1 = load_link_buf(r_nu_o, g, hb + fmu, (M*)nullptr);
            const M m2 = herm(load_link_buf(r_mu_o, g, hb + fnu, (M*)nullptr));
            const M m3 = herm(load_link_buf(r_nu_s, g, hb, (M*)nullptr));
#ifdef QCD_ABL_STAPLE  // diagnostic build: loads kept, products dropped
            add_to(stap, m1); add_to(stap, m2); add_to(stap, m3);
#else
            add_to(stap, mul(mul(m1, m2), m3));
#endif
        }
        {
            // U_nu(x-nu+mu)^+ U_mu(x-nu)^+ U_nu(x-nu)
            const M m3 = load_link_buf(r_nu_o, g, hb + bnu, (M*)nullptr);
            const M m2 = herm(load_link_buf(r_mu_o, g, hb + bnu, (M*)nullptr));
            const M m1 = herm(load_link_buf(r_nu_s, g, hb + bnu + fmu, (M*)nullptr));
#ifdef QCD_ABL_STAPLE
            add_to(stap, m1); add_to(stap, m2); add_to(stap, m3);
#else
            add_to(stap, mul(mul(m1, m2), m3));
#endif
        }
    }
    return stap;
}

// ---------------------------------------------------------------------------------------------
// update pass

#ifdef QCD_TRACE  // diagnostic build: per-wave time stamps of the phases of k_update
__device__ unsigned long long g_trace[8192 * 8];
#define QCD_STAMP(k)                                                                         \
    do {                                                                                     \
        if ((threadIdx.x & 63) == 0) {                                                       \
            const unsigned w_ = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);              \
            if (w_ < 8192) g_trace[w_ * 8 + (k)] = __builtin_readcyclecounter();             \
        }                                                                                    \
    } while (0)
#else
#define QCD_STAMP(k) do {} while (0)
#endif

template <int BLOCK>
__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;  // valid in lane 0 of each wave
}

#ifndef QCD_UPD_WAVES
#define QCD_UPD_WAVES 2   // waves per SIMD the register budget is set for
#endif
template <int GROUP, int DIR, bool GID, int NSKIP, int BLOCK>
__global__ __launch_bounds__(BLOCK, QCD_UPD_WAVES) void k_update(double2* __restrict__ U, const Geom g, const int parity,
                                                  const double beta_eff, const int nhit, const RlxView rv,
                                                  const float* __restrict__ gidr,
                                                  unsigned long long* __restrict__ draws)
{
    __shared__ float rbuf[GID ? 1 : 24 * RLX_WINDOW * BLOCK];
    unsigned long long used = 0;
    Site x;
    int t;  // stream number = the site pair's half index in the slab
#ifdef QCD_STAG_PH  // experiment: de-synchronise the first round of workgroups
    if (blockIdx.x < 1024) {
        const int j = blockIdx.x >> 3;
        const int k = (QCD_STAG_SHIFT < 0 ? ((j * 0x9E3779B1u) >> 20) : (j >> QCD_STAG_SHIFT)) % QCD_STAG_PH;
        for (int i = 0; i < k * QCD_STAG_SLEEPS; i++) __builtin_amdgcn_s_sleep(127);
    }
#endif
    if (site_of_workgroup<BLOCK>(g, blockIdx.x, threadIdx.x, parity, x, t)) {
        const int hb = half_index(g, x) * 16;
        const NbrOff nbr = neighbour_offsets(g, x);
        const __amdgpu_buffer_rsrc_t r_own = link_rsrc<GROUP>(U, g, DIR, parity);
        RlxFast<BLOCK, NSKIP> rs;
        QCD_STAMP(0);
#ifdef QCD_TRACE
        if ((threadIdx.x & 63) == 0) {
            const unsigned w_ = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
            if (w_ < 8192) g_trace[w_ * 8 + 6] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
            if (w_ < 8192) g_trace[w_ * 8 + 7] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));   // XCC_ID
        }
#endif
        mat<GROUP> A = staple<GROUP, DIR>(U, g, hb, nbr, parity);
        QCD_STAMP(1);
        mat<GROUP> Ul = load_link_buf(r_own, g, hb, (mat<GROUP>*)nullptr);
        if (!GID) rs.open(rv, t, rbuf + threadIdx.x);
        QCD_STAMP(2);
        if (GID) {
            // table index: site number inside the slab
            const size_t lg = (size_t)(((x.xs * g.nt + x.t) * g.nz + x.z)) * g.ny + x.y;
            const size_t n = (size_t)g.nx * g.gslice;
            GidSource src;
#pragma unroll
            for (int k = 0; k < 4; k++) src.trial[k] = gidr[k * n + lg];
            src.accept[0] = gidr[4 * n + lg];
            src.accept[1] = gidr[5 * n + lg];
            update_link<true>(Ul, A, beta_eff, nhit, src);
        } else {
            update_link<false>(Ul, A, beta_eff, nhit, rs);
            QCD_STAMP(3);
            int n = rs.close();
            if (n < 0) {
                // window ran dry: repeat this link from its old value with the general stream
                Ul = load_link_buf(r_own, g, hb, (mat<GROUP>*)nullptr);
                RlxStream<BLOCK> slow;
                slow.open(rv, t, rbuf + threadIdx.x);
                update_link<false>(Ul, A, beta_eff, nhit, slow);
                n = slow.close();
            }
            used = (unsigned long long)n;
        }
        store_link_buf(r_own, g, hb, Ul);
        QCD_STAMP(4);
    }
    if (!GID) {
        used = block_sum_u64<BLOCK>(used);
        if ((threadIdx.x & 63) == 0 && used) atomicAdd(draws, used);
    }
}

// ---------------------------------------------------------------------------------------------
// plaquette sums (SU(N)/suncpp/Measurements/Plq.h:46-94): per site the 3 space-space planes
// (xy, xz, yz) and the 3 space-time planes (xt, yt, zt); ReTr[U_mu(x) U_nu(x+mu) U_mu(x+nu)^+ U_nu(x)^+]

template <int GROUP>
__device__ __forceinline__ double retr_plaquette(const double2* __restrict__ U, const Geom& g, const Site& x,
                                                 int mu, int nu)
{
    using M = mat<GROUP>;
    const M m1 = load_site_link<GROUP>(U, g, mu, x);
    const M m2 = load_site_link<GROUP>(U, g, nu, fwd(g, x, mu));
    const M m3 = herm(load_site_link<GROUP>(U, g, mu, fwd(g, x, nu)));
    const M m4 = herm(load_site_link<GROUP>(U, g, nu, x));
    const M p = mul(mul(mul(m1, m2), m3), m4);
    return trace(p).re;
}

template <int BLOCK>
__device__ __forceinline__ void block_reduce_store(double a, double b, double2* __restrict__ partial)
{
    __shared__ double sa[BLOCK / 64], sb[BLOCK / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_down(a, off, 64);
        b += __shfl_down(b, off, 64);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sa[wave] = a;
        sb[wave] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ta = 0.0, tb = 0.0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; w++) {
            ta += sa[w];
            tb += sb[w];
        }
        partial[blockIdx.x] = make_double2(ta, tb);
    }
}

template <int GROUP, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_plaquette(const double2* __restrict__ U, const Geom g,
                                                     double2* __restrict__ partial)
{
    const int t = blockIdx.x * BLOCK + threadIdx.x;
    double spat = 0.0, temp = 0.0;
    if (t < 2 * g.vh_int) {
        const int par = t / g.vh_int;
        const Site x = site_of(g, t - par * g.vh_int, par);
        spat += retr_plaquette<GROUP>(U, g, x, 0, 1);
        spat += retr_plaquette<GROUP>(U, g, x, 0, 2);
        spat += retr_plaquette<GROUP>(U, g, x, 1, 2);
        temp += retr_plaquette<GROUP>(U, g, x, 0, 3);
        temp += retr_plaquette<GROUP>(U, g, x, 1, 3);
        temp += retr_plaquette<GROUP>(U, g, x, 2, 3);
    }
    block_reduce_store<BLOCK>(spat, temp, partial);
}

// sums n double2 partials per output pair in index order (one block): out[2*o], out[2*o+1]
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_reduce(const double2* __restrict__ partial, const int n, const int nout,
                                                  double* __restrict__ out)
{
    __shared__ double sa[BLOCK], sb[BLOCK];
    for (int o = 0; o < nout; o++) {
        double a = 0.0, b = 0.0;
        for (int i = threadIdx.x; i < n; i += BLOCK) {
            const double2 p = partial[(size_t)o * n + i];
            a += p.x;
            b += p.y;
        }
        sa[threadIdx.x] = a;
        sb[threadIdx.x] = b;
        __syncthreads();
        for (int s = BLOCK / 2; s > 0; s >>= 1) {
            if (threadIdx.x < s) {
                sa[threadIdx.x] += sa[threadIdx.x + s];
                sb[threadIdx.x] += sb[threadIdx.x + s];
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            out[2 * o] = sa[0];
            out[2 * o + 1] = sb[0];
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Polyakov loop (SU(N)/suncl/su3cpu.cpp:1383-1433; kernel lattice_polyakov,
// SU(N)/suncl/polyakov.cl:53-141): P = Tr prod_{t=0}^{Nt-1} U_T(x,t), left to right from t = 0.
// partial[0][block] = (sum Re P, sum Im P), partial[1][block] = (sum |P|^2, sum |P|^4)

template <int GROUP, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_polyakov(const double2* __restrict__ U, const Geom g,
                                                    double2* __restrict__ partial, const int nblocks)
{
    const int t = blockIdx.x * BLOCK + threadIdx.x;
    const int n3 = g.nx * g.ny * g.nz;
    double re = 0.0, im = 0.0, p2 = 0.0, p4 = 0.0;
    if (t < n3) {
        Site x;
        x.xs = t / (g.ny * g.nz);
        const int r = t - x.xs * (g.ny * g.nz);
        x.z = r / g.ny;
        x.y = r - x.z * g.ny;
        x.t = 0;
        mat<GROUP> m = load_site_link<GROUP>(U, g, 3, x);
        for (int tt = 1; tt < g.nt; tt++) {
            x.t = tt;
            const mat<GROUP> l = load_site_link<GROUP>(U, g, 3, x);
            m = mul(m, l);
        }
        const cplx tr = trace(m);
        re = tr.re;
        im = tr.im;
        p2 = re * re + im * im;
        p4 = p2 * p2;
    }
    // two independent reductions; the helper owns its LDS, so run it twice through a barrier
    block_reduce_store<BLOCK>(re, im, partial);
    __syncthreads();
    block_reduce_store<BLOCK>(p2, p4, partial + nblocks);
}

// ---------------------------------------------------------------------------------------------
// stand-alone re-unitarisation of every owned link (kernel lattice_GramSchmidt, suncl.cl:349-396)

template <int GROUP, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_reunit(double2* __restrict__ U, const Geom g)
{
    const int t = blockIdx.x * BLOCK + threadIdx.x;
    if (t >= 2 * g.vh_int) return;
    const int par = t / g.vh_int;
    const Site x = site_of(g, t - par * g.vh_int, par);
    const int h = half_index(g, x);
#pragma unroll
    for (int dir = 0; dir < 4; dir++) {
        mat<GROUP> m = load_link(U, g, dir, par, h, (mat<GROUP>*)nullptr);
        gram_schmidt(m);
        store_link(U, g, dir, par, h, m);
    }
}

// cold start: every stored link (halos included) = 1 (kernel lattice_init_cold, suncl.cl:163-187)
template <int GROUP, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_init_cold(double2* __restrict__ U, const Geom g)
{
    const size_t n = (size_t)8 * ncomp<GROUP>::value * g.vh_alloc;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
        const int c = (int)((i / g.vh_alloc) % ncomp<GROUP>::value);
        const bool diag = (GROUP == 3) ? (c == 0 || c == 4) : (c == 0);
        U[i] = make_double2(diag ? 1.0 : 0.0, 0.0);
    }
}

// ---------------------------------------------------------------------------------------------
// reference device layout <-> engine layout.  `table` holds `nslices` x-slices starting at
// stored slice `slice0` (0 = left halo of a slab, or the first owned slice when halo = 0):
// rows of double4, element [lg + row*rowsize], lg = y + z*Ny + t*Ny*Nz + xl*Ny*Nz*Nt.

template <int GROUP, bool TO_ENGINE, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_convert(double2* __restrict__ U, const Geom g, double4* __restrict__ table,
                                                   const int rowsize, const int slice0, const int nslices)
{
    const int lg = blockIdx.x * BLOCK + threadIdx.x;
    if (lg >= nslices * g.gslice) return;
    const int xl = lg / g.gslice;
    int r = lg - xl * g.gslice;
    Site s;
    s.xs = xl + slice0 - g.halo;
    s.t = r / (g.ny * g.nz);
    r -= s.t * (g.ny * g.nz);
    s.z = r / g.ny;
    s.y = r - s.z * g.ny;
    const int par = site_parity(g, s);
    const int h = half_index(g, s);
#pragma unroll
    for (int dir = 0; dir < 4; dir++) {
        if (GROUP == 3) {
            double4* q0 = table + (size_t)(0 * 4 + dir) * rowsize + lg;
            double4* q1 = table + (size_t)(1 * 4 + dir) * rowsize + lg;
            double4* q2 = table + (size_t)(2 * 4 + dir) * rowsize + lg;
            double2* p = U + comp_base<3>(g, dir, par) + h;
            const size_t st = g.vh_alloc;
            if (TO_ENGINE) {
                const double4 a = *q0, b = *q1, c = *q2;
                p[0 * st] = make_double2(a.x, b.x);  // u1
                p[1 * st] = make_double2(a.y, b.y);  // u2
                p[2 * st] = make_double2(a.z, b.z);  // u3
                p[3 * st] = make_double2(c.x, c.z);  // v1
                p[4 * st] = make_double2(c.y, c.w);  // v2
                p[5 * st] = make_double2(a.w, b.w);  // v3
            } else {
                const double2 u1 = p[0 * st], u2 = p[1 * st], u3 = p[2 * st];
                const double2 v1 = p[3 * st], v2 = p[4 * st], v3 = p[5 * st];
                *q0 = make_double4(u1.x, u2.x, u3.x, v3.x);
                *q1 = make_double4(u1.y, u2.y, u3.y, v3.y);
                *q2 = make_double4(v1.x, v2.x, v1.y, v2.y);
            }
        } else {
            double4* q0 = table + (size_t)dir * rowsize + lg;
            double2* p = U + comp_base<2>(g, dir, par) + h;
            const size_t st = g.vh_alloc;
            if (TO_ENGINE) {
                const double4 a = *q0;
                p[0 * st] = make_double2(a.x, a.z);  // u1
                p[1 * st] = make_double2(a.y, a.w);  // u2
            } else {
                const double2 u1 = p[0 * st], u2 = p[1 * st];
                *q0 = make_double4(u1.x, u2.x, u1.y, u2.y);
            }
        }
    }
}

// boundary slice <-> message buffer: buf[c*slice_h + r]
template <int GROUP, bool PACK, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_halo(double2* __restrict__ U, const Geom g, const int dir, const int par,
                                                const int stored_slice, double2* __restrict__ buf)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    constexpr int NC = ncomp<GROUP>::value;
    if (i >= NC * g.slice_h) return;
    const int c = i / g.slice_h;
    const int r = i - c * g.slice_h;
    double2* p = U + comp_base<GROUP>(g, dir, par) + (size_t)c * g.vh_alloc + (size_t)stored_slice * g.slice_h + r;
    if (PACK) buf[i] = *p;
    else *p = buf[i];
}

}  // namespace qcd
