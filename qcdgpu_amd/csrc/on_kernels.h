// on_kernels.h -- HIP kernels of the O(N) scalar path (gfx950): the checkerboard multi-hit Metropolis
// update of the reference's "O(1) radial" model (O(N)/oncl/oncl.cl:74-366, action O(N)/oncl/o1cl.cl:41-144)
// and its measurements (O(N)/oncl/on_measurements_cl.cl:48-214).
//
// One real number per site.  The reference keeps the field in site order; here it is split by
// parity,  phi[parity][h],  h = gid >> 1  (gid = y + z*Ny + t*Ny*Nz + x*Ny*Nz*Nt), so that a pass
// reads its own sites and all four forward neighbours as contiguous runs.  The pass is bound by
// arithmetic (three fp64 logarithms, four fp32 ones, a square root and an exponential per proposal,
// 250 RANLUX steps per site), and what memory traffic there is, is generator state: 200 B per site
// against 4-8 B of field.
//
// Arithmetic types follow the kernel source text literally: its variables are hgpu_float (float
// or double by PRECISION) and its literals are unsuffixed, i.e. double, so a single-precision
// build evaluates most of the action in double and rounds where a value is stored into an
// hgpu_float.  F below is hgpu_float.
#pragma once

#include <hip/hip_runtime.h>

#include "fastmath.h"
#include "kernels.h"
#include "ranlux.h"

// the C++ names of the library live in namespace QCD_NS: qcd, or qcd_f32 in the single-precision build, so that both
// libraries can be loaded into one process and a profile tells their kernels apart
#ifndef QCD_NS
#define QCD_NS qcd
#endif
namespace QCD_NS {

struct OnGeom {
    int nx, ny, nz, nt;
    int nyh;       // ny / 2
    int slice_h;   // half-sites per x slice
    int vh;        // half-sites of the lattice = threads of an update pass
    int idx_shifts;   // nyh, nz and nt all powers of two: their logarithms in bits 0..7, 8..15, 16..; else -1 (divisions)
};

// lattice_parameters[16..22] (O(N)/suncl/suncl.cpp:1987-1994) and SKGROUP
template <typename F>
struct OnPar {
    F z, zeta, eta, b, sqrt_z_lambda_zeta, max_U;
    int skgroup;
};

struct OnSite {
    int x, y, z, t;
};

__device__ __forceinline__ OnSite on_site_of(const OnGeom& g, int h, int par)
{
    OnSite s;
    int yh;
    if (g.idx_shifts >= 0) {   // uniform over the launch
        const int sy = g.idx_shifts & 255, sz = (g.idx_shifts >> 8) & 255, st = g.idx_shifts >> 16;
        yh = h & ((1 << sy) - 1);
        int r = h >> sy;
        s.z = r & ((1 << sz) - 1);
        r >>= sz;
        s.t = r & ((1 << st) - 1);
        s.x = r >> st;
    } else {
        yh = h % g.nyh;
        int r = h / g.nyh;
        s.z = r % g.nz;
        r /= g.nz;
        s.t = r % g.nt;
        s.x = r / g.nt;
    }
    s.y = 2 * yh + ((par + s.x + s.z + s.t) & 1);
    return s;
}
__device__ __forceinline__ int on_half_index(const OnGeom& g, int x, int y, int z, int t)
{
    return ((x * g.nt + t) * g.nz + z) * g.nyh + (y >> 1);
}

// half indices (in the opposite parity's array) of the four forward neighbours, periodic
// (O(N)/kernel/neighbours.cl:123-148)
__device__ __forceinline__ void on_forward(const OnGeom& g, const OnSite& s, int h, int nb[4])
{
    nb[0] = (s.x == g.nx - 1) ? h - (g.nx - 1) * g.slice_h : h + g.slice_h;
    nb[1] = (s.y == g.ny - 1) ? h - (g.nyh - 1) : h + (s.y & 1);
    nb[2] = (s.z == g.nz - 1) ? h - (g.nz - 1) * g.nyh : h + g.nyh;
    nb[3] = (s.t == g.nt - 1) ? h - (g.nt - 1) * g.nyh * g.nz : h + g.nyh * g.nz;
}

// single precision: the hardware's base-2 logarithm (1 ulp) times ln 2 -- the reference's device code calls OpenCL's
// log on a float, which the standard lets be 3 ulp off; the library routine costs twelve times the instructions and the
// action takes four of these per proposal.  A stored field value is the proposal itself, so the last place of a
// logarithm only matters to a proposal that close to its threshold (about one in 1e6).
#ifdef QCD_ON_LIB_LOGF   // diagnostic build: the library logarithm, as before round 3
__device__ __forceinline__ float on_log(float x) { return logf(x); }
#else
__device__ __forceinline__ float on_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
#endif
__device__ __forceinline__ double on_log(double x) { return log(x); }
__device__ __forceinline__ float on_exp(float x) { return expf(x); }
__device__ __forceinline__ double on_exp(double x) { return exp(x); }

// The fp64 logarithms, the square root and the two fp64 quotients of the action come from a Math
// policy (fastmath.h): the update pass uses the table logarithm and the hardware-seeded reciprocal
// and square root (errors of 1e-16 in an action of order one, the size of the difference between
// any two libms); the measurement kernel uses the same.

// o1cl.cl:41-45
template <typename F, class Math>
__device__ __forceinline__ F on_log_one_m(F Ux, const Math& m)
{
    return (F)m.log_of(1.0 - Ux);
}
// o1cl.cl:47-53
template <typename F>
__device__ __forceinline__ F on_phi(F l, F b, F eta)
{
    const F a = (F)((-1.0 + 0.5 * eta * l) * l);
    return (F)(0.5 * a * a * b);
}
template <typename F, class Math>
__device__ __forceinline__ F on_phi_of(F Ux, const OnPar<F>& p, const Math& m)
{
    return on_phi(on_log_one_m(Ux, m), p.b, p.eta);
}

// o1cl.cl:59-144 with forward differences (ON_SUB_SCHEME is commented out, O(N)/kernel/model.cl:60).
// PhiN are the neighbours' Phi values: they do not depend on the proposal, so a site computes
// them once for all its proposals (same inputs, same operations: same bits).
template <typename F, class Math>
__device__ __forceinline__ F on_action(F Ux, const F PhiN[4], const OnPar<F>& p, const Math& m)
{
    const F log_one_m_Ux = on_log_one_m(Ux, m);
    const F Phi = on_phi(log_one_m_Ux, p.b, p.eta);
    F term;
    {
        const F l0 = on_log(PhiN[0] / Phi);
        term = l0 * l0;
        const F l1 = on_log(PhiN[1] / Phi);
        term += l1 * l1;
        const F l2 = on_log(PhiN[2] / Phi);
        term += l2 * l2;
        const F l3 = p.zeta * on_log(PhiN[3] / Phi);
        term += l3 * l3;
    }
    const double skgroupm1_2 = 0.5 * (p.skgroup - 1);
    const double s2 = m.sqrt_of(2.0 * Phi);
    const double mn = (1.9999999999 < s2) ? 1.9999999999 : s2;
    return (F)(-skgroupm1_2 * m.log_of(2.0 - mn) - m.log_of(m.div(1.0 - p.eta * log_one_m_Ux, 1.0 - Ux)) +
               m.div(Phi * (Phi - 1.0 + p.sqrt_z_lambda_zeta * term), p.z));
}

// o1cl.cl:146-154
template <typename F, class Math>
__device__ __forceinline__ F on_physical(F Ux, const OnPar<F>& p, const Math& m)
{
    return (F)m.sqrt_of(2.0 * on_phi_of(Ux, p, m));
}

// Generator state of the O(N) streams: tiles of ON_RLX_TILE streams, a tile's 25 words stored
// one after the other ([tile][word][lane]), so that a workgroup's state is one contiguous 25 KiB
// run per slot.  (Word-major over ALL streams, the link update's layout, puts the 25 words of a
// stream 4*ns bytes apart -- 33 MB at 64^4 -- and every one of a wave's 50 state accesses on a
// page of its own.)
constexpr int ON_RLX_TILE = 256;

struct RlxTileHome {
    unsigned* __restrict__ s0;   // this stream's lane in word 0 of its tile, slot 0
    size_t slot_stride;          // words from slot 0 to slot 1
    int q, rem;

    __device__ __forceinline__ void bind(const RlxView& view, int stream)
    {
        const int tile = stream / ON_RLX_TILE;
        s0 = view.slot0 + ((size_t)tile * ON_RLX_TILE * RLX_PACKED_QUADS + (stream - tile * ON_RLX_TILE)) * 4;
        slot_stride = (size_t)(view.slot1 - view.slot0);
        q = view.q;
        rem = view.rem;
    }
    // packed form in memory (rlx_pack: five 16-byte quads per state, quad q4 of a tile's lane at word
    // ((tile * 5 + q4) * 256 + lane) * 4): the generator state is most of what a pass of this model
    // moves (one 4-byte field value per site against 2 x 80 bytes of state)
    typedef unsigned quad_t __attribute__((ext_vector_type(4)));
    __device__ __forceinline__ void load_state(RlxRegs& r, int slot) const
    {
        const quad_t* __restrict__ p = (const quad_t*)(s0 + (size_t)slot * slot_stride);
        unsigned d[RLX_SLOT_WORDS];
#pragma unroll
        for (int q4 = 0; q4 < RLX_PACKED_QUADS; q4++) {
            const quad_t v = __builtin_nontemporal_load(p + q4 * ON_RLX_TILE);
            d[4 * q4] = v.x; d[4 * q4 + 1] = v.y; d[4 * q4 + 2] = v.z; d[4 * q4 + 3] = v.w;
        }
        rlx_unpack(d, r);
    }
    __device__ __forceinline__ void store_state(const RlxRegs& r, int slot) const
    {
        quad_t* __restrict__ p = (quad_t*)(s0 + (size_t)slot * slot_stride);
        unsigned d[RLX_SLOT_WORDS];
        rlx_pack(r, d);
#pragma unroll
        for (int q4 = 0; q4 < RLX_PACKED_QUADS; q4++) {
            quad_t v;
            v.x = d[4 * q4]; v.y = d[4 * q4 + 1]; v.z = d[4 * q4 + 2]; v.w = d[4 * q4 + 3];
            __builtin_nontemporal_store(v, p + q4 * ON_RLX_TILE);
        }
    }
};

// A thread's RANLUX stream with ONE delivered batch in LDS (column `tid` of a [24][BLOCK] float
// array), refilled on demand.  Same consumption rule as the link update's streams: a pass starts
// at a fresh batch.  Ten proposals need 20 numbers, so a refill only happens for NHIT > 12.
template <int BLOCK, int NSKIP>
struct RlxOne {
    float* buf;
    RlxTileHome h;
    int pos, nb, cur;   // the state in front of batch k is in slot cur ^ (k & 1), k = nb

    __device__ __forceinline__ void skip(RlxRegs& r) const
    {
        if (NSKIP > 0) rlx_skip_static<(NSKIP > 0 ? NSKIP : 24)>(r);
        else rlx_skip(r, h.q, h.rem);
    }
    __device__ __forceinline__ void deliver()
    {
        RlxRegs r;
        h.load_state(r, cur ^ (nb & 1));
        rlx_batch(r, buf, BLOCK);
        skip(r);
        nb++;
        h.store_state(r, cur ^ (nb & 1));
        pos = 0;
    }
    __device__ __forceinline__ void open(const RlxView& view, int stream, float* lds_column)
    {
        h.bind(view, stream);
        buf = lds_column;
        cur = view.cur;   // the slot every stream's state is in for this launch (no per-stream control word)
        nb = 0;
        deliver();
    }
    __device__ __forceinline__ void draw4(float r[4])
    {
        if (pos == 24) deliver();
#pragma unroll
        for (int k = 0; k < 4; k++) r[k] = buf[(pos + k) * BLOCK];
        pos += 4;
    }
    // the stream resumes in front of the batch after the last one delivered; every stream leaves a
    // launch with that state in slot cur ^ 1 (an even number of deliveries, NHIT > 12 only, ends in
    // slot cur: moved over)
    __device__ __forceinline__ void close()
    {
        if ((nb & 1) == 0) move_over();
    }
    __device__ __noinline__ void move_over()   // cold: kept out of line so that it costs the pass no registers
    {
        RlxRegs r;
        h.load_state(r, cur);
        h.store_state(r, cur ^ 1);
    }
};

template <int BLOCK>
__device__ __forceinline__ unsigned long long on_wave_sum(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// update_even / update_odd (oncl.cl:74-219, 221-366): up to NHIT proposals U' = r * max_U, two per
// quad of random numbers, the first with r' < exp(S_old - S_new) is stored.
// wg_counts[2b], [2b+1]: workgroup b's random numbers consumed (4 per quad) and its part of the
// ACC_RATE sum of this pass, sum over sites of NHIT minus the proposals rejected before the accepted one.
template <typename F, int BLOCK, int NSKIP>
__global__ __launch_bounds__(BLOCK) void k_on_update(F* __restrict__ phi, const OnGeom g, const int parity,
                                                     const OnPar<F> p, const int nhit, const RlxView rv,
                                                     unsigned long long* __restrict__ wg_counts)
{
    __shared__ float rbuf[24 * BLOCK];
    __shared__ double2 logtab[LOG_TABLE_ENTRIES];
    MathFast::fill<BLOCK>(logtab);
    __syncthreads();
    MathFast fm;
    fm.tab = logtab;
    const int h = blockIdx.x * BLOCK + threadIdx.x;
    unsigned long long quads = 0, acc = 0;
    if (h < g.vh) {
        F* __restrict__ own = phi + (size_t)parity * g.vh;
        const F* __restrict__ other = phi + (size_t)(parity ^ 1) * g.vh;
        const OnSite s = on_site_of(g, h, parity);
        int nb[4];
        on_forward(g, s, h, nb);
        const F Ux = own[h];
        F PhiN[4];
#pragma unroll
        for (int k = 0; k < 4; k++) PhiN[k] = other[nb[k]];
        RlxOne<BLOCK, NSKIP> rs;
        rs.open(rv, h, rbuf + threadIdx.x);
#pragma unroll
        for (int k = 0; k < 4; k++) PhiN[k] = on_phi_of(PhiN[k], p, fm);
        const F action_old = on_action(Ux, PhiN, p, fm);

        int out = nhit;
        bool flag = false;
        F Unew = Ux;
        const int nhit_2 = nhit >> 1;
        for (int i = 0; i < nhit_2 && !flag; i++) {
            float rnd[4];
            rs.draw4(rnd);
            quads++;
#pragma unroll
            for (int k = 0; k < 2; k++) {
                if (!flag) {
                    const F cand = (F)rnd[2 * k] * p.max_U;
                    const F action_new = on_action(cand, PhiN, p, fm);
                    const F deltaS = on_exp(action_old - action_new);
                    if ((F)rnd[2 * k + 1] < deltaS) {
                        Unew = cand;
                        flag = true;
                    } else {
                        out--;
                    }
                }
            }
        }
        if ((nhit & 1) && !flag) {
            float rnd[4];
            rs.draw4(rnd);
            quads++;
            const F cand = (F)rnd[0] * p.max_U;
            const F action_new = on_action(cand, PhiN, p, fm);
            const F deltaS = on_exp(action_old - action_new);
            if ((F)rnd[1] < deltaS) {
                Unew = cand;
                flag = true;
            } else {
                out--;
            }
        }
        if (flag) own[h] = Unew;
        rs.close();
        acc = (unsigned long long)out;
    }
    // one pair of sums per workgroup, added up by k_on_count (131072 waves adding to one address
    // with atomics take longer than the whole update)
    __shared__ unsigned long long wsum[2][BLOCK / 64];
    quads = on_wave_sum<BLOCK>(quads);
    acc = on_wave_sum<BLOCK>(acc);
    if ((threadIdx.x & 63) == 0) {
        wsum[0][threadIdx.x >> 6] = quads;
        wsum[1][threadIdx.x >> 6] = acc;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        unsigned long long t = 0;
        for (int w = 0; w < BLOCK / 64; w++) t += wsum[threadIdx.x][w];
        wg_counts[2 * (size_t)blockIdx.x + threadIdx.x] = threadIdx.x == 0 ? 4ull * t : t;
    }
}

// sums the per-workgroup pairs of an update pass: counters[0] += numbers drawn, counters[1 + parity] = acceptance sum
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_on_count(const unsigned long long* __restrict__ wg_counts, const int n, const int parity,
                                                    unsigned long long* __restrict__ counters)
{
    __shared__ unsigned long long red[2][BLOCK];
    unsigned long long a = 0, b = 0;
    for (int i = threadIdx.x; i < n; i += BLOCK) {
        a += wg_counts[2 * (size_t)i];
        b += wg_counts[2 * (size_t)i + 1];
    }
    red[0][threadIdx.x] = a;
    red[1][threadIdx.x] = b;
    __syncthreads();
    for (int off = BLOCK / 2; off > 0; off >>= 1) {
        if (threadIdx.x < off) {
            red[0][threadIdx.x] += red[0][threadIdx.x + off];
            red[1][threadIdx.x] += red[1][threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        counters[0] += red[0][0];
        counters[1 + parity] = red[1][0];
    }
}

// lattice_init_cold / lattice_init_hot (oncl.cl:49-72, o1_update_cl.cl:48-60): cold = 0.25; hot =
// r * max_U with the first number of a quad -- the thread of site pair (2h, 2h+1) takes a fresh
// batch and gives the first numbers of its quads 0 and 1 to the two sites
template <typename F, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_on_init(F* __restrict__ phi, const OnGeom g, const int hot, const F max_U,
                                                   const RlxView rv)
{
    __shared__ float rbuf[24 * BLOCK];
    const int h = blockIdx.x * BLOCK + threadIdx.x;
    if (h >= g.vh) return;
    F v0 = (F)0.25, v1 = (F)0.25;
    if (hot) {
        RlxOne<BLOCK, 0> rs;
        rs.open(rv, h, rbuf + threadIdx.x);
        float a[4], b[4];
        rs.draw4(a);
        rs.draw4(b);
        rs.close();
        v0 = (F)a[0] * max_U;
        v1 = (F)b[0] * max_U;
    }
    // site 2h has parity (x + z + t) & 1 (its y is even), site 2h + 1 the other one
    const OnSite s = on_site_of(g, h, 0);
    const int par0 = (s.x + s.z + s.t) & 1;
    phi[(size_t)par0 * g.vh + h] = v0;
    phi[(size_t)(par0 ^ 1) * g.vh + h] = v1;
}

// lattice_measurement, lattice_measurement_plq, lattice_measurement_correlator
// (on_measurements_cl.cl:48-107, 122-154, 167-214) in one pass: per site the action S, the
// physical field, its square, and the field's products with the sites at +step and at
// (+1,+1,+1,+1); fp64 sums, one partial of 5 doubles per workgroup.
template <typename F, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_on_measure(const F* __restrict__ phi, const OnGeom g, const OnPar<F> p,
                                                      const int4 step, double* __restrict__ partial)
{
    __shared__ double red[5][BLOCK / 64];
    __shared__ double2 logtab[LOG_TABLE_ENTRIES];
    MathFast::fill<BLOCK>(logtab);
    __syncthreads();
    const int w = blockIdx.x * BLOCK + threadIdx.x;
    double v[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    if (w < 2 * g.vh) {
        const int par = w / g.vh;
        const int h = w - par * g.vh;
        const F* __restrict__ own = phi + (size_t)par * g.vh;
        const F* __restrict__ other = phi + (size_t)(par ^ 1) * g.vh;
        const OnSite s = on_site_of(g, h, par);
        int nb[4];
        on_forward(g, s, h, nb);
        const F Ux = own[h];
        F PhiN[4];
        MathFast lm;   // as in the update pass (a call at 64^4: 630 -> 380 us); the reference's CPU re-measurement
                       // is reproduced to 1e-12 either way
        lm.tab = logtab;
#pragma unroll
        for (int k = 0; k < 4; k++) PhiN[k] = on_phi_of(other[nb[k]], p, lm);
        v[0] = (double)on_action(Ux, PhiN, p, lm);
        const F f = on_physical(Ux, p, lm);
        v[1] = (double)f;
        v[2] = (double)(F)(f * f);
        // kernel/neighbours.cl:200-213: one conditional subtraction per direction
        int c2[4] = {s.x + step.x, s.y + step.y, s.z + step.z, s.t + step.w};
        int c3[4] = {s.x + 1, s.y + 1, s.z + 1, s.t + 1};
        const int ext[4] = {g.nx, g.ny, g.nz, g.nt};
#pragma unroll
        for (int d = 0; d < 4; d++) {
            if (c2[d] >= ext[d]) c2[d] -= ext[d];
            if (c3[d] >= ext[d]) c3[d] = 0;
        }
        const int p2 = (c2[0] + c2[1] + c2[2] + c2[3]) & 1, p3 = (c3[0] + c3[1] + c3[2] + c3[3]) & 1;
        const F u2 = phi[(size_t)p2 * g.vh + on_half_index(g, c2[0], c2[1], c2[2], c2[3])];
        const F u3 = phi[(size_t)p3 * g.vh + on_half_index(g, c3[0], c3[1], c3[2], c3[3])];
        v[3] = (double)(F)(f * on_physical(u2, p, lm));
        v[4] = (double)(F)(f * on_physical(u3, p, lm));
    }
#pragma unroll
    for (int k = 0; k < 5; k++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off, 64);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        double sum = 0.0;
        for (int wv = 0; wv < BLOCK / 64; wv++) sum += red[threadIdx.x][wv];
        partial[(size_t)blockIdx.x * 5 + threadIdx.x] = sum;
    }
}

// adds the workgroup partials in a fixed order (strided, then a tree); workgroup k does sum k of the five
// (one workgroup looping over the five took 180 us at 64^4)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_on_reduce(const double* __restrict__ partial, const int n, double* __restrict__ out)
{
    __shared__ double red[BLOCK];
    const int k = blockIdx.x;
    double sum = 0.0;
    for (int i = threadIdx.x; i < n; i += BLOCK) sum += partial[(size_t)i * 5 + k];
    red[threadIdx.x] = sum;
    __syncthreads();
    for (int off = BLOCK / 2; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[k] = red[0];
}

// field in site order (the reference's lattice_table, o1_matrix_memory.cl:41-54) <-> phi[parity][h]
template <typename F, bool TO_ENGINE, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_on_convert(F* __restrict__ phi, const OnGeom g, F* __restrict__ table)
{
    const int gid = blockIdx.x * BLOCK + threadIdx.x;
    if (gid >= 2 * g.vh) return;
    const int h = gid >> 1;
    const OnSite s = on_site_of(g, h, 0);
    const int par = ((s.x + s.z + s.t) & 1) ^ (gid & 1);
    if (TO_ENGINE) phi[(size_t)par * g.vh + h] = table[gid];
    else table[gid] = phi[(size_t)par * g.vh + h];
}

}  // namespace QCD_NS
