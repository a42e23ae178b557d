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

// the C++ names of the library live in namespace QCD_NS: qcd, or qcd_f32 in the single-precision build, so that both
// libraries can be loaded into one process and a profile tells their kernels apart
#ifndef QCD_NS
#define QCD_NS qcd
#endif
namespace QCD_NS {

// delivered RANLUX batches held in LDS per thread by the update kernel
constexpr int RLX_WINDOW = 2;
// rows of a thread's LDS column: the window and four more, through which values beyond the window are handed out
constexpr int RLX_ROWS = 24 * RLX_WINDOW + 4;
// the pair kernel's columns have 25 rows more: the generator state in front of the second batch, for streams that go
// on to another site in the same launch (Geom::share)
constexpr int RLX_KEEP_ROW = RLX_ROWS;
constexpr int RLX_ROWS_KEEP = RLX_ROWS + 25;

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
    unsigned* slot0;    // packed generator state in front of a batch: quad q of a stream at word (q * ns + stream) * 4
    unsigned* slot1;    // second copy = slot0 + 20*ns (one allocation); ctl says which of the two is current
    unsigned* ctl;      // (scalar model's engine only) bit 0: slot that holds the state in front of a stream's next batch
    int cur;            // link update: the slot that holds EVERY stream's state in front of its next batch.  A pass
                        // uses each stream once and always leaves its state in the other slot, so the slot is
                        // known per launch and costs no memory access (a per-stream control word did: its load
                        // and store made the 32^4 pass 11 % slower)
    int ns;             // streams
    int q, rem;         // nskip = 24*q + rem
};

// Where a stream's persistent state lives, and the moves both stream types share.
struct RlxHome {
    unsigned* __restrict__ s0;   // this stream's first quad in slot 0; slot 1 lies 20*ns words behind it
    int ns, q, rem;

    __device__ __forceinline__ void bind(const RlxView& view, int stream)
    {
        s0 = view.slot0 + (size_t)4 * stream;
        ns = view.ns;
        q = view.q;
        rem = view.rem;
    }
    // Packed form in memory (rlx_pack): five 16-byte quads per state, quad q at word (q * ns + stream) * 4.
    // Non-temporal: a stream's state is read once and written once per pass, it should not take
    // L2 room from the links.
    typedef unsigned quad_t __attribute__((ext_vector_type(4)));
    __device__ __forceinline__ void load_state(RlxRegs& r, int slot) const
    {
        const quad_t* __restrict__ p = (const quad_t*)(s0 + (size_t)slot * RLX_SLOT_WORDS * ns);
        unsigned d[RLX_SLOT_WORDS];
#pragma unroll
        for (int q4 = 0; q4 < RLX_PACKED_QUADS; q4++) {
            const quad_t v = __builtin_nontemporal_load(p + (size_t)q4 * ns);
            d[4 * q4] = v.x; d[4 * q4 + 1] = v.y; d[4 * q4 + 2] = v.z; d[4 * q4 + 3] = v.w;
        }
        rlx_unpack(d, r);
    }
    __device__ __forceinline__ void store_state(const RlxRegs& r, int slot) const
    {
        quad_t* __restrict__ p = (quad_t*)(s0 + (size_t)slot * RLX_SLOT_WORDS * ns);
        unsigned d[RLX_SLOT_WORDS];
        rlx_pack(r, d);
#pragma unroll
        for (int q4 = 0; q4 < RLX_PACKED_QUADS; q4++) {
            quad_t v;
            v.x = d[4 * q4]; v.y = d[4 * q4 + 1]; v.z = d[4 * q4 + 2]; v.w = d[4 * q4 + 3];
            __builtin_nontemporal_store(v, p + (size_t)q4 * ns);
        }
    }
};

// The stream as the update kernel uses it: two delivered batches (48 values) are written to LDS
// column `tid` when the stream is opened, and the generator's registers are dead after that -- no
// registers held, no generator code in the way while the heat-bath runs.  An SU(3) link update needs
// 18 values plus 4 per rejected trial, so the window covers seven rejections; at beta = 6 one lane
// in 400 touches the second batch and practically none the end of it.
// A lane that does use up the window goes on with the same generator in another form: its state
// (in front of the third batch, rebuilt from the one the opening left in memory) lies in the first
// 24 rows of its LDS column -- the window's values are all spent by then -- and is stepped in a
// run-time loop with run-time indices; every further batch is delivered to rows 24..47 and the
// skip behind it is done at once, so the column always holds the state in front of the next
// batch.  Slow (about a link update's worth of instructions per batch), small, without a call
// and without registers of its own besides a rotation count and the carry; it runs for the few
// lanes that need it, at small beta for many.
// KEEP_LDS: where generate_keep leaves the state in front of the second batch (rows RLX_KEEP_ROW.. of the column, or
// the caller's registers)
template <int BLOCK, int NSKIP, bool KEEP_LDS = false>
struct RlxFast {
    __device__ __forceinline__ void skip(RlxRegs& r) const
    {
        if (NSKIP > 0) rlx_skip_static<(NSKIP > 0 ? NSKIP : 24)>(r);
        else rlx_skip(r, h.q, h.rem);
    }
    float* buf;
    RlxHome h;
    int pos, cur;
    int rot;          // slow form: physical row of logical word k is (k + rot) mod 24
    unsigned carry;   // slow form: the generator's carry
    bool is_kept;     // the state in front of the second batch is in `kept` / the column's keep rows, not in memory
    RlxRegs kept;     // (KEEP_LDS = false) that state

    // The three steps of opening the stream, separate so that the kernel can place the state's
    // loads under other work:
    // attach: bind the stream
    __device__ __forceinline__ void attach(const RlxView& view, int stream, float* lds_column)
    {
        h.bind(view, stream);
        buf = lds_column;
        pos = 0;
        cur = view.cur;
        rot = 0;
        carry = 0u;
        is_kept = false;
    }
    // fetch: request the generator state in front of the stream's next batch
    __device__ __forceinline__ void fetch(RlxRegs& r) const
    {
#if defined(QCD_ABL_RNG_MEM)   // diagnostic build: the generator runs, its state comes from nowhere and goes nowhere
#pragma unroll
        for (int k = 0; k < 24; k++) r.s[k] = (unsigned)(k * 2654435761u + (unsigned)(size_t)buf) & 0xFFFFFFu;
        r.c = 0u;
#elif !defined(QCD_ABL_RNG)
        h.load_state(r, cur);
#endif
    }
    // generate: two delivered batches into LDS, the state between them made persistent
    __device__ __forceinline__ void generate(RlxRegs& r)
    {
#if defined(QCD_ABL_RNG_ALU)   // diagnostic build: the state travels, the generator does not run
        h.store_state(r, cur ^ 1);
#pragma unroll
        for (int k = 0; k < 48; k++) buf[k * BLOCK] = 0.37f + 0.001f * (float)(k % 24);
#elif defined(QCD_ABL_RNG_MEM)
        rlx_batch(r, buf, BLOCK);
        skip(r);
        rlx_batch(r, buf + 24 * BLOCK, BLOCK);
#elif !defined(QCD_ABL_RNG)  // diagnostic build: no generator work, constant numbers
        rlx_batch(r, buf, BLOCK);
        skip(r);
        h.store_state(r, cur ^ 1);   // in front of the second batch: the usual next state
        rlx_batch(r, buf + 24 * BLOCK, BLOCK);
#endif
    }
    // A stream that serves several sites of a launch (Geom::share) goes from one to the next without its state
    // passing through memory.  generate_keep: like generate, but the state in front of the second batch -- where the
    // stream resumes unless the second batch is touched -- is kept instead of stored.
    __device__ __forceinline__ void generate_keep(RlxRegs& r)
    {
        rlx_batch(r, buf, BLOCK);
        skip(r);
        is_kept = true;
        if (KEEP_LDS) {
            unsigned* w = words();
#pragma unroll
            for (int k = 0; k < 24; k++) w[(RLX_KEEP_ROW + k) * BLOCK] = r.s[k];
            w[(RLX_KEEP_ROW + 24) * BLOCK] = r.c;
        } else {
            kept = r;
        }
        rlx_batch(r, buf + 24 * BLOCK, BLOCK);
    }
    // after the update: the state in front of the stream's next batch into r (what close() leaves in memory)
    __device__ __forceinline__ void carry_over(RlxRegs& r)
    {
        const unsigned* w = words();
        if (pos > 24 * RLX_WINDOW) {
            // the slow form keeps the state in front of the next batch in the column's first 24 rows, rotated
#pragma unroll
            for (int k = 0; k < 24; k++) {
                int ph = k + rot;
                ph = ph >= 24 ? ph - 24 : ph;
                r.s[k] = w[ph * BLOCK];
            }
            r.c = carry;
            return;
        }
        if (KEEP_LDS) {
#pragma unroll
            for (int k = 0; k < 24; k++) r.s[k] = w[(RLX_KEEP_ROW + k) * BLOCK];
            r.c = w[(RLX_KEEP_ROW + 24) * BLOCK];
        } else {
            r = kept;
        }
        if (pos > 24) {   // the second batch was touched: the stream resumes in front of the third
            rlx_round(r);
            skip(r);
        }
    }
    // a state from one site to the next through the column's first 25 rows (the window's values are spent by then)
    __device__ __forceinline__ void park(const RlxRegs& r)
    {
        unsigned* w = words();
#pragma unroll
        for (int k = 0; k < 24; k++) w[k * BLOCK] = r.s[k];
        w[24 * BLOCK] = r.c;
    }
    __device__ __forceinline__ void unpark(RlxRegs& r) const
    {
        const unsigned* w = words();
#pragma unroll
        for (int k = 0; k < 24; k++) r.s[k] = w[k * BLOCK];
        r.c = w[24 * BLOCK];
    }
    // the stream for member `sub` of its sites in this launch (`last`: no member follows), and the end of that use
    __device__ __forceinline__ void open_member(const RlxView& view, int stream, float* lds_column, int sub, bool last)
    {
        attach(view, stream, lds_column);
        RlxRegs r;
        if (sub == 0) fetch(r);
        else unpark(r);
        if (last) generate(r);
        else generate_keep(r);
    }
    __device__ __forceinline__ int close_member(bool last)
    {
        if (last) return close();
        RlxRegs r;
        carry_over(r);
        park(r);
        return pos;
    }
    __device__ __forceinline__ void open(const RlxView& view, int stream, float* lds_column)
    {
        attach(view, stream, lds_column);
        RlxRegs r;
        fetch(r);
        generate(r);
    }
    // ---- the slow form
    __device__ __forceinline__ unsigned* words() const { return reinterpret_cast<unsigned*>(buf); }
    // one subtract-with-borrow step on logical words 23 and 9, then a rotation by one word
    // (rlx_step_rotate on the LDS column); returns the new word
    __device__ __forceinline__ unsigned slow_step()
    {
        int ip = rot + 23, jp = rot + 9;
        ip = ip >= 24 ? ip - 24 : ip;
        jp = jp >= 24 ? jp - 24 : jp;
        unsigned* w = words();
        unsigned co;
        const unsigned t = __builtin_subc(w[jp * BLOCK], w[ip * BLOCK], carry, &co);
        carry = co;
        const unsigned v = t & 0xFFFFFFu;
        w[ip * BLOCK] = v;
        rot = rot == 0 ? 23 : rot - 1;
        return v;
    }
    __device__ __forceinline__ void slow_skip(int n)
    {
#pragma unroll 1
        for (int k = 0; k < n; k++) (void)slow_step();
    }
    // the state in front of the second batch (memory, slot cur ^ 1) into the column's first 24 rows
    __device__ __forceinline__ void slow_load()
    {
        typedef unsigned quad_t __attribute__((ext_vector_type(4)));
        unsigned* w = words();
        if (is_kept) {   // a stream that goes on to another site: generate_keep kept that state out of memory
            if (KEEP_LDS) {
#pragma unroll 1
                for (int k = 0; k < 24; k++) w[k * BLOCK] = w[(RLX_KEEP_ROW + k) * BLOCK];
                carry = w[(RLX_KEEP_ROW + 24) * BLOCK];
            } else {
#pragma unroll
                for (int k = 0; k < 24; k++) w[k * BLOCK] = kept.s[k];
                carry = kept.c;
            }
            rot = 0;
            return;
        }
        const quad_t* __restrict__ p = (const quad_t*)(h.s0 + (size_t)(cur ^ 1) * RLX_SLOT_WORDS * h.ns);
        unsigned d[RLX_SLOT_WORDS];
#pragma unroll
        for (int q4 = 0; q4 < RLX_PACKED_QUADS; q4++) {
            const quad_t v = p[(size_t)q4 * h.ns];
            d[4 * q4] = v.x; d[4 * q4 + 1] = v.y; d[4 * q4 + 2] = v.z; d[4 * q4 + 3] = v.w;
        }
#pragma unroll
        for (int j = 0; j < 6; j++) {
            unsigned s4[4];
            rlx_unpack4(d[3 * j], d[3 * j + 1], d[3 * j + 2], s4);
#pragma unroll
            for (int m = 0; m < 4; m++) w[(4 * j + m) * BLOCK] = s4[m];
        }
        carry = d[18];
        rot = 0;
    }
    __device__ __forceinline__ void slow_store()
    {
        typedef unsigned quad_t __attribute__((ext_vector_type(4)));
        quad_t* __restrict__ p = (quad_t*)(h.s0 + (size_t)(cur ^ 1) * RLX_SLOT_WORDS * h.ns);
        const unsigned* w = words();
        unsigned d[RLX_SLOT_WORDS];
#pragma unroll
        for (int j = 0; j < 6; j++) {
            unsigned s4[4];
#pragma unroll
            for (int m = 0; m < 4; m++) {
                int ph = 4 * j + m + rot;
                ph = ph >= 24 ? ph - 24 : ph;
                s4[m] = w[ph * BLOCK];
            }
            rlx_pack4(s4, d[3 * j], d[3 * j + 1], d[3 * j + 2]);
        }
        d[18] = carry;
        d[19] = 0u;
#pragma unroll
        for (int q4 = 0; q4 < RLX_PACKED_QUADS; q4++) {
            quad_t v;
            v.x = d[4 * q4]; v.y = d[4 * q4 + 1]; v.z = d[4 * q4 + 2]; v.w = d[4 * q4 + 3];
            p[(size_t)q4 * h.ns] = v;
        }
    }
    // deliver the next batch to rows 24..47 and skip behind it (rlx_batch + rlx_skip, run-time form;
    // the word the tiny-value correction reads is word (32 - k) mod 24 of the frame the batch started in)
    __device__ __forceinline__ void slow_refill()
    {
        const int nskip = NSKIP > 0 ? NSKIP : 24 * h.q + h.rem;
        if (pos == 24 * RLX_WINDOW) {
            slow_load();
            slow_skip(24 + nskip);   // over the second batch, which the opening delivered, and the skip behind it
        }
        const int rot0 = rot;
        float* out = buf + 24 * BLOCK;
        const unsigned* w = words();
#pragma unroll 1
        for (int k = 0; k < 24; k++) {
            const unsigned ui = slow_step();
            float uni = (float)ui * RLX_TWOM24;
            if (ui < 4096u) {
                int ph = ((32 - k) % 24) + rot0;
                ph = ph >= 24 ? ph - 24 : ph;
                uni = uni + RLX_TWOM24 * ((float)w[ph * BLOCK] * RLX_TWOM24);
                if (uni == 0.0f) uni = RLX_TWOM24 * RLX_TWOM24;
            }
            out[k * BLOCK] = uni;
        }
        slow_skip(nskip);
    }
    // N values (4 or 2).  Inside the window: N reads.  Beyond: one at a time, through the N rows behind
    // the window, so that the refill's code exists once per call site.
    template <int N>
    __device__ __forceinline__ void draw(float r[N])
    {
#ifdef QCD_ABL_RNG
#pragma unroll
        for (int k = 0; k < N; k++) r[k] = 0.37f + 0.001f * (float)(pos++ % 24);
        return;
#endif
        if (__builtin_expect(pos + N <= 24 * RLX_WINDOW, 1)) {
#pragma unroll
            for (int k = 0; k < N; k++) r[k] = buf[(pos + k) * BLOCK];
            pos += N;
            return;
        }
#pragma unroll 1
        for (int k = 0; k < N; k++) {
            float v;
            if (pos < 24 * RLX_WINDOW) {
                v = buf[pos * BLOCK];
            } else {
                const int m = pos % 24;
                if (m == 0) slow_refill();
                v = buf[(24 + m) * BLOCK];
            }
            buf[(24 * RLX_WINDOW + k) * BLOCK] = v;
            pos++;
        }
#pragma unroll
        for (int k = 0; k < N; k++) r[k] = buf[(24 * RLX_WINDOW + k) * BLOCK];
    }
    __device__ __forceinline__ void draw4(float r[4]) { draw<4>(r); }
    __device__ __forceinline__ void draw2(float r[2]) { draw<2>(r); }
    // number of values used; leaves the stream's state in front of its next batch in slot cur ^ 1
    __device__ __forceinline__ int close()
    {
#if !defined(QCD_ABL_RNG) && !defined(QCD_ABL_RNG_MEM) && !defined(QCD_ABL_RNG_ALU)
        if (pos > 24 * RLX_WINDOW) {
            slow_store();   // the slow form keeps the state in front of the next batch
        } else if (pos > 24) {
            // rare (two or more rejected trials): the second batch was touched, the stream resumes in
            // front of the third; rebuild that state from the one in front of the second, in place
            // (every stream leaves a pass with its state in slot cur ^ 1)
            RlxRegs r;
            h.load_state(r, cur ^ 1);
            rlx_round(r);
            skip(r);
            h.store_state(r, cur ^ 1);
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
__device__ __forceinline__ double cos_2pi_of(float r, double) { return cospi(2.0 * (double)r); }
__device__ __forceinline__ float cos_2pi_of(float r, float) { return cospif(2.0f * r); }
__device__ __forceinline__ real cos_2pi(float r) { return cos_2pi_of(r, real()); }
__device__ __forceinline__ void sincos_2pi(float r, double& s, double& c) { sincospi(2.0 * (double)r, &s, &c); }
__device__ __forceinline__ void sincos_2pi(float r, float& s, float& c) { sincospif(2.0f * r, &s, &c); }

// `m`: where reciprocal, square root and logarithm come from (fastmath.h).  beta_inv = 1 / beta.
template <bool GID, class Rng, class Math>
__device__ __forceinline__ bool heatbath_su2(const su2& stap, real beta, real beta_inv, int nhit, Rng& rng,
                                             const Math& m, su2& out)
{
    const real Mx = (stap.e[0][0].re + stap.e[1][1].re);
    const real My = (stap.e[0][0].im - stap.e[1][1].im);
    const real Mz = (stap.e[0][1].re - stap.e[1][0].re);
    const real Mw = (stap.e[0][1].im + stap.e[1][0].im);

    real det, idet;
    m.sqrt_rcp(Mx * Mx + My * My + Mz * Mz + Mw * Mw, det, idet);
    const real bdet = beta * det;
    const real ibdet = beta_inv * idet;   // the reference divides by bdet; <= 1 ulp apart

    su2 aH;
    aH.e[0][0] = mk(Mx * idet, -My * idet);
    aH.e[0][1] = mk(-Mz * idet, -Mw * idet);
    aH.e[1][1] = mk(aH.e[0][0].re, -aH.e[0][0].im);
    aH.e[1][0] = mk(-aH.e[0][1].re, aH.e[0][1].im);

    bool flag = false;
    real delta = 0;
    float rnd[4];
    if (GID) {
        // INTS = 2: the first trial is always accepted (update_su2.cpp:87-90)
        rng.draw4(rnd);
        const real cosrnd = cos_2pi(rnd[1]);
        delta = cosrnd * cosrnd * ibdet;
        flag = true;
    } else {
        if constexpr (REAL_IS_DOUBLE) {
            const float ibdet32 = m.rcp32(bdet);
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
                    const double cosrnd = m.cos_2pi(rnd[1], double());
                    const double d = -(m.log_of(1.0 - rnd[0]) + cosrnd * cosrnd * m.log_of(1.0 - rnd[2])) * ibdet;
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
                const double cosrnd = m.cos_2pi(a1, double());
                delta = -(m.log_of(1.0 - a0) + cosrnd * cosrnd * m.log_of(1.0 - a2)) * ibdet;
            }
        } else {
            // single precision: the test as the reference's float build evaluates it (hgpu_float arithmetic,
            // SU(N)/suncl/su3_update_cl.cl:597-649), with the hardware logarithm
            int i = 0;
            while ((i < nhit) && !flag) {
                rng.draw4(rnd);
                const float cosrnd = m.cos_2pi(rnd[1], float());
                const float d = -(m.log_of(1.0f - rnd[0]) + cosrnd * cosrnd * m.log_of(1.0f - rnd[2])) * ibdet;
                if (rnd[3] * rnd[3] <= (1.0f - 0.5f * d)) {
                    flag = true;
                    delta = d;
                }
                i++;
            }
        }
    }
    if (flag) {
        rng.draw2(rnd);
        const real cosal = (real)1 - delta;
        const real costh = (real)2 * rnd[0] - (real)1;
        const real sinth = m.sqrt_of((real)1 - costh * costh);
        const real sinal = m.sqrt_of((real)1 - cosal * cosal);
        real sinphi, cosphi;
        m.sincos_2pi(rnd[1], sinphi, cosphi);
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
template <int P, int Q, bool GID, class Rng, class Math>
__device__ __forceinline__ void cm_step(su3& U, const su3& A, real beta, real beta_inv, int nhit, Rng& rng,
                                        const Math& m)
{
    su2 blk;
    blk.e[0][0] = dot3(U.e[P][0], U.e[P][1], U.e[P][2], A.e[0][P], A.e[1][P], A.e[2][P]);
    blk.e[0][1] = dot3(U.e[P][0], U.e[P][1], U.e[P][2], A.e[0][Q], A.e[1][Q], A.e[2][Q]);
    blk.e[1][0] = dot3(U.e[Q][0], U.e[Q][1], U.e[Q][2], A.e[0][P], A.e[1][P], A.e[2][P]);
    blk.e[1][1] = dot3(U.e[Q][0], U.e[Q][1], U.e[Q][2], A.e[0][Q], A.e[1][Q], A.e[2][Q]);
    su2 r;
    if (heatbath_su2<GID>(blk, beta, beta_inv, nhit, rng, m, r)) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const cplx up = U.e[P][j], uq = U.e[Q][j];
            U.e[P][j] = dot2(r.e[0][0], r.e[0][1], up, uq);
            U.e[Q][j] = dot2(r.e[1][0], r.e[1][1], up, uq);
        }
    }
}

template <bool GID, class Rng, class Math>
__device__ __forceinline__ void update_link(su3& U, const su3& A, real beta, real beta_inv, int nhit, Rng& rng,
                                            const Math& m)
{
#ifdef QCD_ABL_HB  // diagnostic build: no heat-bath, keeps loads, staple products and the store
    add_to(U, A);
    gram_schmidt(U, m);
    return;
#endif
    cm_step<0, 1, GID>(U, A, beta, beta_inv, nhit, rng, m);
    cm_step<0, 2, GID>(U, A, beta, beta_inv, nhit, rng, m);
    cm_step<1, 2, GID>(U, A, beta, beta_inv, nhit, rng, m);
    gram_schmidt(U, m);
}
// SU(2): the new link does not depend on the old one; unchanged when no trial is accepted
// (update_su2.cpp:121-125 with final = 1, SU(N)/suncpp/Update/sun_update.h:94)
template <bool GID, class Rng, class Math>
__device__ __forceinline__ void update_link(su2& U, const su2& A, real beta, real beta_inv, int nhit, Rng& rng,
                                            const Math& m)
{
    su2 r;
    if (heatbath_su2<GID>(A, beta, beta_inv, nhit, rng, m, r)) U = r;
}

// ---------------------------------------------------------------------------------------------
// Overrelaxation (microcanonical move).  The reference has none: this is the heat-bath's own
// construction with the random element replaced by a reflection.  There the new SU(2) element is
// c * aH, aH = (M/det)^+ (SU(N)/suncpp/su2/update_su2.cpp:56-66); here c = aH, the element
// aH * aH, which leaves Re Tr of the 2x2 block of U*A -- hence the action -- unchanged.
template <class Math>
__device__ __forceinline__ su2 overrelax_su2(const su2& blk, const Math& m)
{
    const real Mx = (blk.e[0][0].re + blk.e[1][1].re);
    const real My = (blk.e[0][0].im - blk.e[1][1].im);
    const real Mz = (blk.e[0][1].re - blk.e[1][0].re);
    const real Mw = (blk.e[0][1].im + blk.e[1][0].im);
    real det, idet;
    m.sqrt_rcp(Mx * Mx + My * My + Mz * Mz + Mw * Mw, det, idet);
    su2 aH;
    aH.e[0][0] = mk(Mx * idet, -My * idet);
    aH.e[0][1] = mk(-Mz * idet, -Mw * idet);
    aH.e[1][1] = mk(aH.e[0][0].re, -aH.e[0][0].im);
    aH.e[1][0] = mk(-aH.e[0][1].re, aH.e[0][1].im);
    return mul(aH, aH);
}
template <int P, int Q, class Math>
__device__ __forceinline__ void cm_overrelax_step(su3& U, const su3& A, const Math& m)
{
    su2 blk;
    blk.e[0][0] = dot3(U.e[P][0], U.e[P][1], U.e[P][2], A.e[0][P], A.e[1][P], A.e[2][P]);
    blk.e[0][1] = dot3(U.e[P][0], U.e[P][1], U.e[P][2], A.e[0][Q], A.e[1][Q], A.e[2][Q]);
    blk.e[1][0] = dot3(U.e[Q][0], U.e[Q][1], U.e[Q][2], A.e[0][P], A.e[1][P], A.e[2][P]);
    blk.e[1][1] = dot3(U.e[Q][0], U.e[Q][1], U.e[Q][2], A.e[0][Q], A.e[1][Q], A.e[2][Q]);
    const su2 r = overrelax_su2(blk, m);
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const cplx up = U.e[P][j], uq = U.e[Q][j];
        U.e[P][j] = dot2(r.e[0][0], r.e[0][1], up, uq);
        U.e[Q][j] = dot2(r.e[1][0], r.e[1][1], up, uq);
    }
}
template <class Math>
__device__ __forceinline__ void overrelax_link(su3& U, const su3& A, const Math& m)
{
    cm_overrelax_step<0, 1>(U, A, m);
    cm_overrelax_step<0, 2>(U, A, m);
    cm_overrelax_step<1, 2>(U, A, m);
    gram_schmidt(U, m);
}
template <class Math>
__device__ __forceinline__ void overrelax_link(su2& U, const su2& A, const Math& m)
{
    const su2 r = overrelax_su2(mul(U, A), m);
    U = mul(r, U);
}

// ---------------------------------------------------------------------------------------------
// staple sum (SU(N)/suncpp/Update/sun_update.h:43-77): nu ascending, forward then backward

// hb = 16 * half index of x, par = parity of x (wave-uniform)
template <int GROUP, int DIR>
__device__ __forceinline__ mat<GROUP> staple(const real2* __restrict__ U, const Geom& g, const int hb,
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

// The same staple sum for lattices with an odd extent: every neighbour's parity is looked up from its own
// coordinates (across the wrap of an odd direction it is the site's own parity), plain loads.
template <int GROUP, int DIR>
__device__ __forceinline__ mat<GROUP> staple_any(const real2* __restrict__ U, const Geom& g, const Site& x)
{
    using M = mat<GROUP>;
    M stap = zero<GROUP>();
    const Site xmu = fwd(g, x, DIR);
#pragma unroll
    for (int nu = 0; nu < 4; nu++) {
        if (nu == DIR) continue;
        {
            const M m1 = load_site_link<GROUP>(U, g, nu, xmu);
            const M m2 = herm(load_site_link<GROUP>(U, g, DIR, fwd(g, x, nu)));
            const M m3 = herm(load_site_link<GROUP>(U, g, nu, x));
            add_to(stap, mul(mul(m1, m2), m3));
        }
        {
            const Site xb = bwd(g, x, nu);
            const M m1 = herm(load_site_link<GROUP>(U, g, nu, fwd(g, xb, DIR)));
            const M m2 = herm(load_site_link<GROUP>(U, g, DIR, xb));
            const M m3 = load_site_link<GROUP>(U, g, nu, xb);
            add_to(stap, mul(mul(m1, m2), m3));
        }
    }
    return stap;
}
// Phase of a site in a pass over the links of direction `dir` on a lattice with odd extents.  Across the wrap of
// an odd direction nu the staple of x reads U_dir(x +- nu) on a site of x's own parity, which the same pass
// updates.  The reference's sequential program (SU(N)/suncpp/Update/sun_update.h:79-117) visits the sites in
// index order, so the site at coordinate N_nu - 1 reads the NEW link of the site at 0 and that one the OLD link
// of the site at N_nu - 1.  Sites are therefore updated in phases by the number of odd directions nu != dir in which
// they sit at N_nu - 1: every same-parity link a site reads belongs to the phase before (new) or after (old) its own.
__device__ __forceinline__ int update_phase(const Geom& g, const Site& x, int dir)
{
    int k = 0;
    if (dir != 0 && (g.odd & 1) && x.xs == g.nx - 1) k++;
    if (dir != 2 && (g.odd & 4) && x.z == g.nz - 1) k++;
    if (dir != 3 && (g.odd & 8) && x.t == g.nt - 1) k++;
    return k;
}

// SU(3) staple sum, software-pipelined and with a small register footprint.  Every staple term
// is itself an SU(3) matrix, so only its first two rows are formed by products (from two rows of
// the leading factor and the full other factor) and the third row is (row0 x row1)*, the rule the
// stored links are completed by; rounding differs from full 3x3 products in the last bit only.
// The loads of a term are requested while the products of the one before are computed, so a
// wave's memory latency hides behind its own arithmetic and not only behind the other wave of
// the SIMD.
struct rows2 {
    cplx e[2][3];
};
__device__ __forceinline__ cplx dot3c(cplx a1, cplx a2, cplx a3, cplx b1, cplx b2, cplx b3)
{
    return dot3(a1, a2, a3, conj(b1), conj(b2), conj(b3));
}

struct raw_link {
    real2 c[6];
};
#ifndef QCD_NU_AUX
#define QCD_NU_AUX 0
#endif
template <int AUX = 0>
__device__ __forceinline__ raw_link issue_link(__amdgpu_buffer_rsrc_t r, const Geom& g, int byte_off)
{
    raw_link l;
#pragma unroll
    for (int c = 0; c < 6; c++)
        l.c[c] = buf_load_c<AUX>(r, byte_off, c * g.vh_alloc * CB);
    return l;
}
__device__ __forceinline__ rows2 as_rows2(const raw_link& l)
{
    rows2 m;
#pragma unroll
    for (int c = 0; c < 6; c++) m.e[c / 3][c % 3] = mk(l.c[c].x, l.c[c].y);
    return m;
}
__device__ __forceinline__ su3 as_su3(const raw_link& l)
{
    su3 m;
#pragma unroll
    for (int c = 0; c < 6; c++) m.e[c / 3][c % 3] = mk(l.c[c].x, l.c[c].y);
    reconstruct(m);
    return m;
}
// Ordering pins for the pipeline below.  The instruction selector is free to sink arithmetic
// below later loads (and then everything is in flight at once and spills); an empty volatile asm
// that takes the values as in/out operands and clobbers memory fixes the order: loads cannot
// cross it, and arithmetic on the pinned values cannot start before / finish after it.
__device__ __forceinline__ void pin(raw_link& l)
{
    asm volatile("" : "+v"(l.c[0].x), "+v"(l.c[0].y), "+v"(l.c[1].x), "+v"(l.c[1].y), "+v"(l.c[2].x), "+v"(l.c[2].y),
                      "+v"(l.c[3].x), "+v"(l.c[3].y), "+v"(l.c[4].x), "+v"(l.c[4].y), "+v"(l.c[5].x), "+v"(l.c[5].y)
                 :: "memory");
}
__device__ __forceinline__ void pin(su3& m)
{
    asm volatile("" : "+v"(m.e[0][0].re), "+v"(m.e[0][0].im), "+v"(m.e[0][1].re), "+v"(m.e[0][1].im),
                      "+v"(m.e[0][2].re), "+v"(m.e[0][2].im), "+v"(m.e[1][0].re), "+v"(m.e[1][0].im),
                      "+v"(m.e[1][1].re), "+v"(m.e[1][1].im), "+v"(m.e[1][2].re), "+v"(m.e[1][2].im),
                      "+v"(m.e[2][0].re), "+v"(m.e[2][0].im), "+v"(m.e[2][1].re), "+v"(m.e[2][1].im),
                      "+v"(m.e[2][2].re), "+v"(m.e[2][2].im)
                 :: "memory");
}
// term 2k: forward staple in the k-th direction nu != DIR, term 2k+1: the backward one.
// The first two links of term k+1 are requested before the second product of term k, its third
// link before the first product of term k+1.
struct raw_pair {
    raw_link a, b;
};
__device__ __forceinline__ void pin(raw_pair& t)
{
    pin(t.a);
    pin(t.b);
}
__device__ __forceinline__ void pin(rows2& m)
{
    asm volatile("" : "+v"(m.e[0][0].re), "+v"(m.e[0][0].im), "+v"(m.e[0][1].re), "+v"(m.e[0][1].im),
                      "+v"(m.e[0][2].re), "+v"(m.e[0][2].im), "+v"(m.e[1][0].re), "+v"(m.e[1][0].im),
                      "+v"(m.e[1][1].re), "+v"(m.e[1][1].im), "+v"(m.e[1][2].re), "+v"(m.e[1][2].im)
                 :: "memory");
}
// Cache policy of the staple's requests.  A non-temporal load tells the L2 that its lines will not be
// asked for again, and the L2 acts on it: with the hint on all 19 requests of a link update a 32^4 pass
// takes 185 us instead of 118.  A link is asked for by up to six sites of a pass, so the hint belongs on the
// request that comes last, and only there.  Measured one request at a time (tools/abl_dirs.sh):
//   U_nu(x-nu)    the closing link of a backward term; the line was first asked for as U_nu(x'+mu) by
//                 x' = x-nu-mu, one term earlier in the pipeline of a block that runs earlier or beside
//                 this one: hint on, -3.0 us per pass
//   U_mu(x)       the link that is updated is read once: hint on, -1.6 us
//   U_nu(x), U_nu(x-nu+mu)  the same line for x and x' = x+nu-mu; which of the two comes later depends on
//                 how the blocks run: hint on either costs 1.5-5 us
//   U_mu(x+-nu)   six requests per line from up to five blocks: no request is reliably the last (+-0.5 us)
// (SU(2), whose lattice stays in the memory-side cache: the hint on U_nu(x-nu) makes a 32^4 pass 2 us slower.)
// One mask per direction mu: bit 3 * which + k asks for the non-temporal policy on request `which` of the
// k-th direction nu != mu (which: 0 = U_nu(x+mu), 1 = U_mu(x+nu), 2 = U_nu(x), 3 = U_mu(x-nu),
// 4 = U_nu(x-nu+mu), 5 = U_nu(x-nu)); bit 18: the link that is updated.
#ifndef QCD_NT_D0
#define QCD_NT_D0 0x78000
#define QCD_NT_D1 0x78000
#define QCD_NT_D2 0x78000
#define QCD_NT_D3 0x78000
#endif
constexpr int AUX_NT = 2;
__host__ __device__ constexpr unsigned nt_mask(int mu)
{
    return mu == 0 ? (QCD_NT_D0) : (mu == 1 ? (QCD_NT_D1) : (mu == 2 ? (QCD_NT_D2) : (QCD_NT_D3)));
}
__host__ __device__ constexpr int staple_aux(int mu, int nu, int which)
{
    const int k = nu < mu ? nu : nu - 1;
    return ((nt_mask(mu) >> (3 * which + k)) & 1u) ? AUX_NT : ((which == 1 || which == 3) ? 0 : QCD_NU_AUX);
}
#ifdef QCD_ABL_DROP   // diagnostic build: the requests named by the mask (bit = `which` above) are not made (unit link instead)
template <int WHICH, int AUX>
__device__ __forceinline__ raw_link issue_or_drop(__amdgpu_buffer_rsrc_t r, const Geom& g, int byte_off)
{
    if (((QCD_ABL_DROP) >> WHICH) & 1) {
        raw_link l;
#pragma unroll
        for (int c = 0; c < 6; c++) {
            l.c[c].x = (c == 0 || c == 4) ? real(1) : real(0);
            l.c[c].y = real(0);
        }
        return l;
    }
    return issue_link<AUX>(r, g, byte_off);
}
#else
template <int WHICH, int AUX>
__device__ __forceinline__ raw_link issue_or_drop(__amdgpu_buffer_rsrc_t r, const Geom& g, int byte_off)
{
    return issue_link<AUX>(r, g, byte_off);
}
#endif
#ifdef QCD_ABL_MU_SAME   // diagnostic build: the six requests for U_mu(x +- nu) all go to the block's own index
#define QCD_MU_OFF(o) 0
#else
#define QCD_MU_OFF(o) (o)
#endif
template <int DIR, int TERM>
__device__ __forceinline__ raw_pair issue_ab(const real2* __restrict__ U, const Geom& g, const int hb,
                                             const NbrOff& n, const int par)
{
    constexpr int k = TERM / 2;
    constexpr int nu = (k < DIR) ? k : k + 1;
    const int fmu = off_fwd(n, DIR);
    raw_pair t;
    if (TERM % 2 == 0) {
        t.a = issue_or_drop<0, staple_aux(DIR, nu, 0)>(link_rsrc<3>(U, g, nu, par ^ 1), g, hb + fmu);
        t.b = issue_or_drop<1, staple_aux(DIR, nu, 1)>(link_rsrc<3>(U, g, DIR, par ^ 1), g, hb + QCD_MU_OFF(off_fwd(n, nu)));
    } else {
        const int bnu = off_bwd(n, nu);
        t.a = issue_or_drop<3, staple_aux(DIR, nu, 3)>(link_rsrc<3>(U, g, DIR, par ^ 1), g, hb + QCD_MU_OFF(bnu));
        t.b = issue_or_drop<4, staple_aux(DIR, nu, 4)>(link_rsrc<3>(U, g, nu, par), g, hb + bnu + fmu);
    }
    return t;
}
template <int DIR, int TERM>
__device__ __forceinline__ raw_link issue_c(const real2* __restrict__ U, const Geom& g, const int hb,
                                            const NbrOff& n, const int par)
{
    constexpr int k = TERM / 2;
    constexpr int nu = (k < DIR) ? k : k + 1;
    if (TERM % 2 == 0) return issue_or_drop<2, staple_aux(DIR, nu, 2)>(link_rsrc<3>(U, g, nu, par), g, hb);
    return issue_or_drop<5, staple_aux(DIR, nu, 5)>(link_rsrc<3>(U, g, nu, par ^ 1), g, hb + off_bwd(n, nu));
}
// first product of a term: two rows of a * b^+ (forward) or of a * b (backward)
template <int TERM>
__device__ __forceinline__ rows2 first_product(const raw_pair& t)
{
    const rows2 a = as_rows2(t.a);
    const su3 b = as_su3(t.b);
    rows2 p;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            p.e[i][j] = (TERM % 2 == 0) ? dot3c(a.e[i][0], a.e[i][1], a.e[i][2], b.e[j][0], b.e[j][1], b.e[j][2])
                                        : dot3(a.e[i][0], a.e[i][1], a.e[i][2], b.e[0][j], b.e[1][j], b.e[2][j]);
    return p;
}
// second product and accumulation: p * c^+ (forward), p^+ * c (backward, p completed first)
template <int TERM>
__device__ __forceinline__ void second_product(su3& stap, const rows2& p2, const raw_link& cl)
{
    const su3 c = as_su3(cl);
    su3 r;
    if (TERM % 2 == 0) {
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 3; j++)
                r.e[i][j] = dot3c(p2.e[i][0], p2.e[i][1], p2.e[i][2], c.e[j][0], c.e[j][1], c.e[j][2]);
    } else {
        su3 p;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) p.e[i][j] = p2.e[i][j];
        reconstruct(p);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 3; j++)
                r.e[i][j] = dot3(conj(p.e[0][i]), conj(p.e[1][i]), conj(p.e[2][i]), c.e[0][j], c.e[1][j], c.e[2][j]);
    }
    reconstruct(r);
    add_to(stap, r);
}
struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};
// the pipeline's state.  Diagnostic build QCD_PAIR_MU: `an` holds the leading link of the NEXT
// backward term, U_mu(x-nu), requested right behind the forward term's U_mu(x+nu), which lies in the
// same (nu = y) or partly the same (nu = z) cache lines, so that the second request meets the first
// one's pending miss in the L1.  Measured at 32^4: 6.4 % fewer L1->L2 read requests, the same number
// fewer L2 hits, L2 misses and the pass time unchanged -- the pass is bound behind the L2, not by
// the request path in front of it.
struct StaplePipe {
    su3 stap;
    raw_pair ab;
    raw_link c;
#ifdef QCD_PAIR_MU
    raw_link an;
#endif
};
template <int DIR, int TERM>
__device__ __forceinline__ void issue_next_ab(StaplePipe& sp, const real2* __restrict__ U, const Geom& g, const int hb,
                                              const NbrOff& n, const int par)
{
#ifdef QCD_PAIR_MU
    constexpr int k = TERM / 2;
    constexpr int nu = (k < DIR) ? k : k + 1;
    constexpr bool paired = (QCD_PAIR_MU == 1) || nu == 1 || nu == 2;
    if (!paired) {
        sp.ab = issue_ab<DIR, TERM>(U, g, hb, n, par);
    } else if (TERM % 2 == 0) {
        sp.ab = issue_ab<DIR, TERM>(U, g, hb, n, par);
        sp.an = issue_link(link_rsrc<3>(U, g, DIR, par ^ 1), g, hb + off_bwd(n, nu));
    } else {
        sp.ab.a = sp.an;
        sp.ab.b = issue_link<QCD_NU_AUX>(link_rsrc<3>(U, g, nu, par), g, hb + off_bwd(n, nu) + off_fwd(n, DIR));
    }
#else
    sp.ab = issue_ab<DIR, TERM>(U, g, hb, n, par);
#endif
}
// `between` runs after the first product, where the next term's first loads are requested (the
// last term has none: the caller's own requests take their place)
template <int DIR, int TERM, class Hook = NoHook>
__device__ __forceinline__ void pipe_step(StaplePipe& sp, const real2* __restrict__ U,
                                           const Geom& g, const int hb, const NbrOff& n, const int par,
                                           Hook between = Hook())
{
    pin(sp.ab);
    rows2 p = first_product<TERM>(sp.ab);
    pin(p);
    if (TERM < 5) issue_next_ab<DIR, (TERM < 5 ? TERM + 1 : 5)>(sp, U, g, hb, n, par);
    between();
    pin(sp.c);
    second_product<TERM>(sp.stap, p, sp.c);
    pin(sp.stap);
    if (TERM < 5) sp.c = issue_c<DIR, (TERM < 5 ? TERM + 1 : 5)>(U, g, hb, n, par);
}
// the pipeline in two parts, so that the caller can request what it needs next (its own link,
// the random stream's state) before the last term's arithmetic
template <int DIR>
__device__ __forceinline__ void staple_pipe_head(StaplePipe& sp, const real2* __restrict__ U, const Geom& g,
                                                 const int hb, const NbrOff& n, const int par)
{
    sp.stap = zero<3>();
    issue_next_ab<DIR, 0>(sp, U, g, hb, n, par);
    sp.c = issue_c<DIR, 0>(U, g, hb, n, par);
    pipe_step<DIR, 0>(sp, U, g, hb, n, par);
    pipe_step<DIR, 1>(sp, U, g, hb, n, par);
    pipe_step<DIR, 2>(sp, U, g, hb, n, par);
    pipe_step<DIR, 3>(sp, U, g, hb, n, par);
    pipe_step<DIR, 4>(sp, U, g, hb, n, par);
}
template <int DIR, class Hook>
__device__ __forceinline__ void staple_pipe_tail(StaplePipe& sp, const real2* __restrict__ U, const Geom& g,
                                                 const int hb, const NbrOff& n, const int par, Hook between)
{
    pipe_step<DIR, 5>(sp, U, g, hb, n, par, between);
}

// ---------------------------------------------------------------------------------------------
// update pass

#ifdef QCD_TRACE  // diagnostic build: per-wave time stamps of the phases of k_update
__device__ unsigned long long g_trace[8192 * 8];
#define QCD_STAMP(k)                                                                         \
    do {                                                                                     \
        if ((threadIdx.x & 63) == 0) {                                                       \
            const unsigned w_ = wb * (BLOCK / 64) + (threadIdx.x >> 6);              \
            if (w_ < 8192) g_trace[w_ * 8 + (k)] = wall_clock64();             \
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
// issue priority of a wave while it gathers the staple (mostly waiting for memory) and while it
// generates random numbers and runs the heat-bath: the arithmetic phase goes first, so that the
// wave sharing the SIMD spends its waiting time under it (measured: -1.5 % per pass)
#ifndef QCD_PRIO_MEM
#define QCD_PRIO_MEM 0
#define QCD_PRIO_ALU 3
#endif
// everything an update launch needs, as ONE kernel argument: the kernel re-reads it from the
// kernel-argument segment at the start of every trip of its loop (scalar loads), so that no
// argument and nothing derived from one has to stay in registers from one site to the next
// the count of random numbers drawn is kept as DRAW_SLOTS partial counts (element k * DRAW_SLOT_STRIDE);
// their sum is the count
constexpr int DRAW_SLOTS = 64;
constexpr int DRAW_SLOT_STRIDE = 16;   // 128 bytes

struct UpdArgs {
    real2* U;
    Geom g;
    int parity;
    int nhit;
    double beta_eff;
    double beta_inv;       // 1 / beta_eff
    RlxView rv;
    const float* gidr;
    unsigned long long* draws;
    int nblocks;   // blocks of BLOCK sites in the pass
    int ngroups;   // nblocks / g.share: groups of blocks whose sites share their random streams, block after block;
                   // workgroup b takes the groups b, b + gridDim.x, ...
    int phase;     // lattices with an odd extent: only the sites of this phase are updated (update_phase); else -1
    int reverse;   // pair kernel: walk the x-slices downwards (every other pass: the end of the pass before is what the
                   // memory-side cache still holds)
    // Edge signal of a slab pass (edge_local = 0: none).  The first edge_local blocks of every
    // region are the slices next to the slab's two faces (the launch starts there); when the last
    // of them is stored and visible to the device, edge_seq is written to *edge_flag, on which
    // the stream that packs the boundary slices waits while this launch goes on with the interior.
    int edge_local;
    unsigned edge_total;     // edge blocks of the launch, all regions
    unsigned edge_seq;
    unsigned* edge_count;    // device memory, zero between launches
    unsigned* edge_flag;     // signal memory (hipMallocSignalMemory)
};

// called by a whole wavefront after the stores of an edge block
__device__ __forceinline__ void edge_block_done(const UpdArgs& a, const int lane)
{
    // The links of an edge block were stored with write-through (STORE_AUX_THROUGH): when the
    // wavefront's stores are acknowledged they are in memory, where the packing kernel -- on any
    // XCD, starting with an L2 that holds nothing of them -- finds them.  (A release fence at device
    // scope would do as well but writes back the whole L2 of the XCD each time: 512 of them made a
    // 32^4 pass 40 us longer.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
        const unsigned done = __hip_atomic_fetch_add(a.edge_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (done == a.edge_total) {
            __hip_atomic_store(a.edge_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.edge_flag, a.edge_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ANY: a lattice with an odd extent (staple_any, phases)
template <int GROUP, int DIR, bool GID, int NSKIP, int BLOCK, bool ANY = false>
__global__ __launch_bounds__(BLOCK, QCD_UPD_WAVES) void k_update(const UpdArgs args_by_value)
{
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass needs the signature only
    __shared__ float rbuf[GID ? 1 : RLX_ROWS * BLOCK];
    __shared__ double2 logtab[LOG_TABLE_ENTRIES];
    __shared__ double2 trigtab[REAL_IS_DOUBLE ? TRIG_TABLE_ENTRIES : 1];   // the fp64 accept step's cosine and sine (MathFast::sincos_2pi)
    MathFast::fill<BLOCK>(logtab);
    if constexpr (REAL_IS_DOUBLE) MathFast::fill_trig<BLOCK>(trigtab);
    __syncthreads();
    MathFast fm;
    fm.tab = logtab;
    fm.trig = trigtab;
    unsigned long long used = 0;
    (void)args_by_value;
    typedef const UpdArgs __attribute__((address_space(4))) * ArgPtr;
    int wg = blockIdx.x, sub = 0;   // group of blocks, member of the group
#pragma unroll 1
    for (;;) {
    ArgPtr ap = (ArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ap));   // a fresh, opaque pointer every trip: nothing read through it is carried over
    if (wg >= ap->ngroups) break;
    real2* __restrict__ U = ap->U;
    Geom g;   // member by member: a whole-struct copy out of the argument segment goes through scratch
    g.nx = ap->g.nx; g.ny = ap->g.ny; g.nz = ap->g.nz; g.nt = ap->g.nt;
    g.halo = ap->g.halo; g.x0 = ap->g.x0; g.nyh = ap->g.nyh; g.slice_h = ap->g.slice_h;
    g.vh_int = ap->g.vh_int; g.vh_alloc = ap->g.vh_alloc; g.gsites = ap->g.gsites; g.gslice = ap->g.gslice;
    g.nzr = ap->g.nzr; g.ntr = ap->g.ntr; g.rz = ap->g.rz; g.rt = ap->g.rt; g.region_h = ap->g.region_h; g.idx_shifts = ap->g.idx_shifts;
    g.tz = ap->g.tz; g.tt = ap->g.tt; g.tx = ap->g.tx;
    g.ux0 = ap->g.ux0; g.unx = ap->g.unx; g.odd = ap->g.odd; g.share = ap->g.share; g.tq = ap->g.tq;
    const int wb = ((wg / (g.nzr * g.ntr)) * g.share + sub) * (g.nzr * g.ntr) + wg % (g.nzr * g.ntr);
    const bool last = sub == g.share - 1;
    if (ANY) { g.tw[0] = ap->g.tw[0]; g.tw[1] = ap->g.tw[1]; g.tw[2] = ap->g.tw[2]; g.tw[3] = ap->g.tw[3]; }
    const int parity = ap->parity;
    const int nhit = ap->nhit;
    const real beta_eff = (real)ap->beta_eff;
    RlxView rv;
    rv.slot0 = ap->rv.slot0; rv.slot1 = ap->rv.slot1; rv.ctl = nullptr; rv.cur = ap->rv.cur;
    rv.ns = ap->rv.ns; rv.q = ap->rv.q; rv.rem = ap->rv.rem;
    const float* __restrict__ gidr = ap->gidr;
    Site x;
    int t;  // stream number = the site pair's half index in the slab
    if (site_of_workgroup<BLOCK>(g, wb, threadIdx.x, parity, x, t) && (!ANY || update_phase(g, x, DIR) == ap->phase)) {
        const int hb = half_index(g, x) * CB;
#ifdef QCD_ABL_LOCAL  // diagnostic build: every neighbour is the site itself (minimal traffic, wrong physics)
        NbrOff nbr;
        nbr.f0 = nbr.f1 = nbr.f2 = nbr.f3 = nbr.b0 = nbr.b1 = nbr.b2 = nbr.b3 = 0;
#else
        const NbrOff nbr = neighbour_offsets(g, x);
#endif
        const __amdgpu_buffer_rsrc_t r_own = link_rsrc<GROUP>(U, g, DIR, parity);
        RlxFast<BLOCK, NSKIP> rs;
        QCD_STAMP(0);
#ifdef QCD_TRACE
        if ((threadIdx.x & 63) == 0) {
            const unsigned w_ = wb * (BLOCK / 64) + (threadIdx.x >> 6);
            if (w_ < 8192) g_trace[w_ * 8 + 6] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
            if (w_ < 8192) g_trace[w_ * 8 + 7] = __builtin_readcyclecounter();   // shader-clock stamp at start
        }
#endif
        mat<GROUP> A, Ul;
        if constexpr (ANY) {
            A = staple_any<GROUP, DIR>(U, g, x);
            Ul = load_site_link<GROUP>(U, g, DIR, x);   // twisted like every link that is read, stored as it comes out (suncl.cl:525-536)
            if (!GID) rs.open_member(rv, t, rbuf + threadIdx.x, sub, last);
        } else if constexpr (GROUP == 3) {
            // the stream's state is requested as soon as the staple's registers are free (requesting
            // state and link under the last staple term costs 40 registers and gains nothing)
            if (!GID) rs.attach(rv, t, rbuf + threadIdx.x);
            __builtin_amdgcn_s_setprio(QCD_PRIO_MEM);
            StaplePipe sp;
            staple_pipe_head<DIR>(sp, U, g, hb, nbr, parity);
            raw_link own;
            RlxRegs r;
            staple_pipe_tail<DIR>(sp, U, g, hb, nbr, parity, NoHook());
            own = issue_link<(((nt_mask(DIR) >> 18) & 1u) ? AUX_NT : 0)>(r_own, g, hb);
            if (!GID) {
                if (sub == 0) rs.fetch(r);
                else rs.unpark(r);
            }
            A = sp.stap;
            QCD_STAMP(1);
            pin(own);
            Ul = as_su3(own);
            __builtin_amdgcn_s_setprio(QCD_PRIO_ALU);
            if (!GID) {
                if (last) rs.generate(r);
                else rs.generate_keep(r);
            }
        } else {
            A = staple<GROUP, DIR>(U, g, hb, nbr, parity);
            QCD_STAMP(1);
            Ul = load_link_buf(r_own, g, hb, (mat<GROUP>*)nullptr);
            if (!GID) rs.open_member(rv, t, rbuf + threadIdx.x, sub, last);
        }
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
            update_link<true>(Ul, A, beta_eff, (real)ap->beta_inv, nhit, src, fm);
        } else {
            update_link<false>(Ul, A, beta_eff, (real)ap->beta_inv, nhit, rs, fm);
            QCD_STAMP(3);
            used += (unsigned long long)rs.close_member(last);
            store_link_buf(r_own, g, hb, Ul);
        }
        if (GID) store_link_buf(r_own, g, hb, Ul);
        QCD_STAMP(4);
#ifdef QCD_TRACE
        if ((threadIdx.x & 63) == 0) {
            const unsigned w_ = wb * (BLOCK / 64) + (threadIdx.x >> 6);
            if (w_ < 8192) g_trace[w_ * 8 + 5] = __builtin_readcyclecounter();   // shader-clock stamp at end
        }
#endif
    }
    if (last) {
        sub = 0;
        wg += gridDim.x;
    } else {
        sub++;
    }
    }
    if (!GID) {
        used = block_sum_u64<BLOCK>(used);
        // one of DRAW_SLOTS counters, a cache line apart: atomics on one address are done one after
        // the other in the L2, and every wave of the launch arrives here at about the same time
        if ((threadIdx.x & 63) == 0 && used) atomicAdd(args_by_value.draws + (size_t)(blockIdx.x % DRAW_SLOTS) * DRAW_SLOT_STRIDE, used);
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// where a thread's site is: byte offset of its half index, neighbour offsets, random stream
struct SiteRef {
    int hb;
    NbrOff nbr;
    int stream;
};
template <int BLOCK>
__device__ __forceinline__ SiteRef site_ref(const Geom& g, int wb, int tid, int parity)
{
    Site x;
    int t = 0;
    (void)site_of_workgroup<BLOCK>(g, wb, tid, parity, x, t);   // every block of the launch is full
    SiteRef r;
    r.hb = half_index(g, x) * CB;
    r.nbr = neighbour_offsets(g, x);
    r.stream = t;
    return r;
}

// ---------------------------------------------------------------------------------------------
// The SU(3) heat-bath pass as a two-stage pipeline inside the workgroup ("gather" and "heat-bath"
// wavefronts).  One workgroup of eight wavefronts per CU, four of them gathering staple sums and
// four running the heat-bath, in TEAMS teams: a team works on ONE block of 64 sites at a time --
// its gather wavefronts each form the partial staple sum of their share of the six terms and
// hand it over through LDS, its heat-bath wavefronts take turns: generate the random numbers,
// add the partial sums, run the heat-bath, store the links -- while the gather side is already at
// the team's next block.
//
// Why.  The two halves of a link update want different things: the gather waits for memory and
// needs registers for links in flight, the heat-bath is a long chain of dependent fp64 arithmetic
// with the staple, the link and its temporaries live.  Split over two wavefronts of one SIMD, each
// SIMD always has one instruction stream that mostly waits and one that mostly computes (two whole
// updates side by side are in the same phase as often as not), a block's requests come in a
// shorter burst, and the gather side can run ahead of the heat-bath side across block boundaries.
// Measured at 32^4, beta 6: 124.6 us per pass against 127.9 us for the kernel with one wavefront per
// block, L2 misses 5.5 M against 5.76 M per launch.  The pass stays bound by the bytes its L2 misses
// move (5.6 TB/s): neither side waits for the other (sleeps per block: gather 0, heat-bath 3).
// Variants measured and dropped: teams of 2 + 2 or 4 + 4 wavefronts sharing one block (TEAMS 2, 1:
// 127 and 134 us, no fewer misses), the requests of a direction issued pairwise per array (152 us),
// a cap on the gather wavefronts with loads in flight (3 of 4: 130 us, 2: 137 us).
//
// Roles follow the SIMD a wavefront finds itself on (HW_ID): with more than 168 registers per lane
// a SIMD holds two wavefronts, so the eight of a workgroup are two per SIMD whatever the order they
// were placed in; the first to arrive gathers, the second runs the heat-bath, so every SIMD has one
// stream that mostly waits for memory and one that mostly computes.
//
// The partial sums of a team's gather wavefronts are added in wavefront order, the terms inside a
// partial sum in the reference's order (nu ascending, forward then backward).
#ifdef QCD_GH_STATS   // diagnostic build: sleeps spent waiting for the other side, [0] gather side, [1] heat-bath side
__device__ unsigned long long g_gh_spins[2];
#define QCD_GH_SPIN(k) spins++
#define QCD_GH_SPIN_DECL unsigned long long spins = 0
#define QCD_GH_SPIN_FLUSH(k) do { if (lane == 0 && spins) atomicAdd(&g_gh_spins[k], spins); } while (0)
#else
#define QCD_GH_SPIN(k) do {} while (0)
#define QCD_GH_SPIN_DECL do {} while (0)
#define QCD_GH_SPIN_FLUSH(k) do {} while (0)
#endif
constexpr int GH_SIDE = 4;      // gather wavefronts per workgroup = heat-bath wavefronts per workgroup
#ifndef QCD_GH_TEAMS
#define QCD_GH_TEAMS 4          // teams per workgroup: 4 (pairs), 2 or 1; measured at 32^4: 124, 127, 134 us per pass
#endif
constexpr int GH_TEAMS = QCD_GH_TEAMS;
constexpr int GH_SLOTS = 2;     // hand-over slots per team
template <int TEAMS>
struct GhShared {
    static constexpr int PER = GH_SIDE / TEAMS;             // wavefronts of either kind per team
    real2 stap[TEAMS][GH_SLOTS][PER][9][64];              // partial staple sums: 18 reals per lane
    float rbuf[GH_SIDE][RLX_ROWS_KEEP * 64];         // random numbers of the heat-bath wavefronts, and a kept generator state
    double2 logtab[LOG_TABLE_ENTRIES];
    double2 trigtab[REAL_IS_DOUBLE ? TRIG_TABLE_ENTRIES : 1];   // the fp64 accept step's cosine and sine (MathFast::sincos_2pi)
    unsigned full_seq[TEAMS][GH_SLOTS][PER];                // gather wavefront j has put block (full_seq - 1) into the slot
    unsigned free_seq[TEAMS][GH_SLOTS];                     // the heat-bath side has read block (free_seq - 1) out of the slot
    unsigned simd_count[4];
#ifdef QCD_GH_SYNC
    unsigned gsync;                                         // experimental: arrivals of the gather wavefronts at their common steps
#endif
};
#ifdef QCD_GH_SYNC   // experimental build: the gather wavefronts of a workgroup start every block (1) or every term (2) together
__device__ __forceinline__ void gh_sync(unsigned* ctr, unsigned& epoch, const int lane)
{
    epoch++;
    if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned)GH_SIDE * epoch) __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ void gh_sync_leave(unsigned* ctr, const int lane)
{
    if (lane == 0) __hip_atomic_fetch_add(ctr, 1u << 24, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
#endif

// the team's it-th block of sites: workgroup b works for region b % regions, the teams of the
// workgroups of a region take consecutive groups of its blocks (the `share` blocks whose sites share their random
// streams), one member after the other; < 0: no more blocks
// reverse: the groups of a region taken from the last x-slice down to the first (see UpdArgs::reverse); the order
// inside a group is the streams' order of use and stays
template <int TEAMS>
__device__ __forceinline__ int gh_block(const Geom& g, const int nblocks, const int team, const int it, const int reverse = 0)
{
    const int regions = g.nzr * g.ntr;
    const int wgs = (int)gridDim.x / regions;
    const int grp_it = it / g.share;
    const int ngroups = nblocks / regions / g.share;
    int local = (grp_it * wgs + (int)blockIdx.x / regions) * TEAMS + team;
    if ((int)blockIdx.x >= wgs * regions || local >= ngroups) return -1;
    if (reverse) local = ngroups - 1 - local;
    return (local * g.share + (it - grp_it * g.share)) * regions + (int)blockIdx.x % regions;
}

// gather side: terms T0 .. T0 + NT - 1 of every block of the team, as one software pipeline that
// runs across the block boundaries (the loads of a block's first term are requested before the
// second product of the block before)
template <int DIR, int T0, int NT, int TEAMS>
__device__ __forceinline__ void gh_gather(GhShared<TEAMS>& sh, const UpdArgs& a, const Geom& g, const int team, const int j,
                                          const int lane)
{
    QCD_GH_SPIN_DECL;
    const int parity = a.parity;
    int wb = gh_block<TEAMS>(g, a.nblocks, team, 0, a.reverse);
#ifdef QCD_GH_SYNC
    unsigned epoch = 0;
    if (wb < 0) gh_sync_leave(&sh.gsync, lane);
#endif
    if (wb < 0) return;
    SiteRef cur = site_ref<64>(g, wb, lane, parity);
    raw_pair ab = issue_ab<DIR, T0>(a.U, g, cur.hb, cur.nbr, parity);
    raw_link cl = issue_c<DIR, T0>(a.U, g, cur.hb, cur.nbr, parity);
#pragma unroll 1
    for (int it = 0;; it++) {
        const int wbn = gh_block<TEAMS>(g, a.nblocks, team, it + 1, a.reverse);
        const bool has_next = wbn >= 0;
        SiteRef nxt = cur;
        if (has_next) nxt = site_ref<64>(g, wbn, lane, parity);
        su3 stap = zero<3>();
#if defined(QCD_GH_SYNC) && QCD_GH_SYNC == 1
        gh_sync(&sh.gsync, epoch, lane);
#endif
#if defined(QCD_GH_SYNC) && QCD_GH_SYNC == 2
#define QCD_GH_TERM_SYNC gh_sync(&sh.gsync, epoch, lane);
#else
#define QCD_GH_TERM_SYNC
#endif
#define QCD_GH_TERM(K)                                                                                   \
        if (K < NT) {                                                                                    \
            constexpr int T = T0 + (K < NT ? K : 0), TN = T0 + (K + 1 < NT ? K + 1 : 0);                 \
            pin(ab);                                                                                     \
            rows2 p = first_product<T>(ab);                                                              \
            pin(p);                                                                                      \
            QCD_GH_TERM_SYNC                                                                             \
            if (K + 1 < NT) ab = issue_ab<DIR, TN>(a.U, g, cur.hb, cur.nbr, parity);                     \
            else if (has_next) ab = issue_ab<DIR, T0>(a.U, g, nxt.hb, nxt.nbr, parity);                  \
            pin(cl);                                                                                     \
            second_product<T>(stap, p, cl);                                                              \
            pin(stap);                                                                                   \
            if (K + 1 < NT) cl = issue_c<DIR, TN>(a.U, g, cur.hb, cur.nbr, parity);                      \
            else if (has_next) cl = issue_c<DIR, T0>(a.U, g, nxt.hb, nxt.nbr, parity);                   \
        }
        QCD_GH_TERM(0) QCD_GH_TERM(1) QCD_GH_TERM(2) QCD_GH_TERM(3) QCD_GH_TERM(4) QCD_GH_TERM(5)
#undef QCD_GH_TERM
#ifdef QCD_ABL_FUSE_LOADS   // diagnostic build: the x and z passes also request what the y and t staples of their sites need (values unused)
        if constexpr (DIR == 0 || DIR == 2) {
            constexpr int D2 = DIR + 1;
            real acc = 0;
#define QCD_GH_EXTRA(T)                                                                                  \
            {                                                                                            \
                raw_pair e = issue_ab<D2, T>(a.U, g, cur.hb, cur.nbr, parity);                           \
                raw_link f = issue_c<D2, T>(a.U, g, cur.hb, cur.nbr, parity);                            \
                pin(e);                                                                                  \
                pin(f);                                                                                  \
                _Pragma("unroll") for (int c = 0; c < 6; c++)                                            \
                    acc += e.a.c[c].x + e.a.c[c].y + e.b.c[c].x + e.b.c[c].y + f.c[c].x + f.c[c].y;      \
                asm volatile("" : "+v"(acc));                                                            \
            }
            QCD_GH_EXTRA(0) QCD_GH_EXTRA(1) QCD_GH_EXTRA(2) QCD_GH_EXTRA(3) QCD_GH_EXTRA(4) QCD_GH_EXTRA(5)
#undef QCD_GH_EXTRA
            stap.e[0][0].re += acc * (real)0;
        }
#endif
        const int s = it % GH_SLOTS;
        if (it >= GH_SLOTS) {
            // the slot is free once the heat-bath side has read block it - GH_SLOTS out of it
            while (__hip_atomic_load(&sh.free_seq[team][s], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned)(it - GH_SLOTS + 1)) {
                __builtin_amdgcn_s_sleep(4);
                QCD_GH_SPIN(0);
            }
        }
        real2* dst = &sh.stap[team][s][j][0][lane];
#pragma unroll
        for (int k = 0; k < 9; k++) dst[k * 64] = make_real2(stap.e[k / 3][k % 3].re, stap.e[k / 3][k % 3].im);
        __hip_atomic_store(&sh.full_seq[team][s][j], (unsigned)(it + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (!has_next) break;
        cur = nxt;
    }
#ifdef QCD_GH_SYNC
    gh_sync_leave(&sh.gsync, lane);
#endif
    QCD_GH_SPIN_FLUSH(0);
}

// OVR: the overrelaxation pass through the same pipeline -- the second wavefront of a pair reflects the link
// (overrelax_link) instead of drawing a new one: no generator, no random numbers, nothing counted
template <int DIR, int NSKIP, int TEAMS, bool OVR = false>
__global__ __launch_bounds__(512, 2) void k_update_gh(const UpdArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    using Sh = GhShared<TEAMS>;
    constexpr int PER = Sh::PER;
    __shared__ Sh sh;
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 4) sh.simd_count[threadIdx.x] = 0u;
#ifdef QCD_GH_SYNC
    if (threadIdx.x == 0) sh.gsync = 0u;
#endif
    if (threadIdx.x < TEAMS * GH_SLOTS * PER) (&sh.full_seq[0][0][0])[threadIdx.x] = 0u;
    if (threadIdx.x < TEAMS * GH_SLOTS) (&sh.free_seq[0][0])[threadIdx.x] = 0u;
    if constexpr (!OVR) MathFast::fill<512>(sh.logtab);   // the reflection takes no logarithm
    if constexpr (!OVR && REAL_IS_DOUBLE) MathFast::fill_trig<512>(sh.trigtab);
    __syncthreads();
    const unsigned simd = (__builtin_amdgcn_s_getreg((4 << 0) | (4 << 6) | (1 << 11))) & 3u;   // HW_ID bits 5:4
    unsigned role = 0;
    if (lane == 0) role = atomicAdd(&sh.simd_count[simd], 1u);
    role = (unsigned)__builtin_amdgcn_readfirstlane((int)role);
    int member = (int)simd;   // 0 .. 3 among the wavefronts of its kind
    __syncthreads();
    if (sh.simd_count[0] != 2u || sh.simd_count[1] != 2u || sh.simd_count[2] != 2u || sh.simd_count[3] != 2u) {
        // not two wavefronts on every SIMD (not expected): roles by wavefront number instead
        role = threadIdx.x >> 8;
        member = (threadIdx.x >> 6) & 3;
    }
    const int team = member / PER, j = member % PER;
    const Geom g = a.g;
    if (role == 0) {
        __builtin_amdgcn_s_setprio(QCD_PRIO_MEM);
        if (PER == 1) gh_gather<DIR, 0, 6, TEAMS>(sh, a, g, team, j, lane);
        else if (PER == 2) {
            if (j == 0) gh_gather<DIR, 0, 3, TEAMS>(sh, a, g, team, j, lane);
            else gh_gather<DIR, 3, 3, TEAMS>(sh, a, g, team, j, lane);
        } else {
            // four gather wavefronts: 2 + 2 + 1 + 1 terms
            if (j == 0) gh_gather<DIR, 0, 2, TEAMS>(sh, a, g, team, j, lane);
            else if (j == 1) gh_gather<DIR, 2, 2, TEAMS>(sh, a, g, team, j, lane);
            else if (j == 2) gh_gather<DIR, 4, 1, TEAMS>(sh, a, g, team, j, lane);
            else gh_gather<DIR, 5, 1, TEAMS>(sh, a, g, team, j, lane);
        }
    } else {
        // ---- heat-bath side: this wavefront takes the team's blocks j, j + PER, ...
        const int parity = a.parity;
        MathFast fm;
        fm.tab = sh.logtab;
        fm.trig = sh.trigtab;
        unsigned long long used = 0;
        QCD_GH_SPIN_DECL;
        float* rcol = &sh.rbuf[member][lane];
        __builtin_amdgcn_s_setprio(QCD_PRIO_ALU);
        RlxRegs r;   // the generator state: in registers from one site of a stream to the next (share > 1 needs PER = 1)
#pragma unroll 1
        for (int it = j;; it += PER) {
            const int wb = gh_block<TEAMS>(g, a.nblocks, team, it, a.reverse);
            if (wb < 0) break;
            const SiteRef cur = site_ref<64>(g, wb, lane, parity);
            const __amdgpu_buffer_rsrc_t r_own = link_rsrc<3>(a.U, g, DIR, parity);
            if constexpr (OVR) {
                raw_link own = issue_link<(((nt_mask(DIR) >> 18) & 1u) ? AUX_NT : 0)>(r_own, g, cur.hb);
                const int s = it % GH_SLOTS;
                while (__hip_atomic_load(&sh.full_seq[team][s][0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned)(it + 1)) {
                    __builtin_amdgcn_s_sleep(4);
                    QCD_GH_SPIN(1);
                }
                static_assert(!OVR || PER == 1, "the overrelaxation pass runs as pairs of wavefronts");
                const real2* src = &sh.stap[team][s][0][0][lane];
                su3 A;
#pragma unroll
                for (int k = 0; k < 9; k++) {
                    const real2 v = src[k * 64];
                    A.e[k / 3][k % 3] = mk(v.x, v.y);
                }
                pin(A);
                __hip_atomic_store(&sh.free_seq[team][s], (unsigned)(it + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                pin(own);
                su3 Ul = as_su3(own);
                overrelax_link(Ul, A, fm);
                store_link_buf(r_own, g, cur.hb, Ul);
                continue;
            }
            RlxFast<64, NSKIP, true> rs;
            rs.attach(a.rv, cur.stream, rcol);
            raw_link own = issue_link<(((nt_mask(DIR) >> 18) & 1u) ? AUX_NT : 0)>(r_own, g, cur.hb);
            const int sub = it % g.share;
            const bool last = sub == g.share - 1;
            if (sub == 0) rs.fetch(r);
            if (last) rs.generate(r);
            else rs.generate_keep(r);
            const int s = it % GH_SLOTS;
            su3 A = zero<3>();
#pragma unroll
            for (int m = 0; m < PER; m++) {
                while (__hip_atomic_load(&sh.full_seq[team][s][m], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned)(it + 1)) {
                    __builtin_amdgcn_s_sleep(4);
                    QCD_GH_SPIN(1);
                }
                const real2* src = &sh.stap[team][s][m][0][lane];
                su3 P;
#pragma unroll
                for (int k = 0; k < 9; k++) {
                    const real2 v = src[k * 64];
                    P.e[k / 3][k % 3] = mk(v.x, v.y);
                }
                if (m == 0) A = P;
                else add_to(A, P);
            }
            pin(A);
            __hip_atomic_store(&sh.free_seq[team][s], (unsigned)(it + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            pin(own);
            su3 Ul = as_su3(own);
            update_link<false>(Ul, A, (real)a.beta_eff, (real)a.beta_inv, a.nhit, rs, fm);
            if (last) {
                used += (unsigned long long)rs.close();
                // the stream is done with this launch: nothing of r goes on to the next block, which fetches
#pragma unroll
                for (int k = 0; k < 24; k++) asm volatile("" : "=v"(r.s[k]));
                asm volatile("" : "=v"(r.c));
            } else {
                used += (unsigned long long)rs.pos;
                rs.carry_over(r);
            }
            if (a.edge_local > 0 && wb / (g.nzr * g.ntr) < a.edge_local) {
                store_link_buf<STORE_AUX_THROUGH>(r_own, g, cur.hb, Ul);
                edge_block_done(a, lane);
            } else {
                store_link_buf(r_own, g, cur.hb, Ul);
            }
        }
        used = block_sum_u64<64>(used);
        if (lane == 0 && used) atomicAdd(a.draws + (size_t)((blockIdx.x * GH_SIDE + member) % DRAW_SLOTS) * DRAW_SLOT_STRIDE, used);
        QCD_GH_SPIN_FLUSH(1);
    }
#endif
}

// overrelaxation pass over the links of direction DIR on the sites of one parity: same gather
// as the heat-bath pass, no random numbers
template <int GROUP, int DIR, int BLOCK, bool ANY = false>
__global__ __launch_bounds__(BLOCK, 2) void k_overrelax(real2* __restrict__ U, const Geom g, const int parity, const int phase)
{
    Site x;
    int t;
    if (!site_of_workgroup<BLOCK>(g, blockIdx.x, threadIdx.x, parity, x, t)) return;
    if (ANY && update_phase(g, x, DIR) != phase) return;
    const int hb = half_index(g, x) * CB;
    const NbrOff nbr = neighbour_offsets(g, x);
    const __amdgpu_buffer_rsrc_t r_own = link_rsrc<GROUP>(U, g, DIR, parity);
    MathFast fm;
    fm.tab = nullptr;   // no logarithm on this path
    mat<GROUP> A;
    if constexpr (ANY) {
        A = staple_any<GROUP, DIR>(U, g, x);
    } else if constexpr (GROUP == 3) {
        StaplePipe sp;
        staple_pipe_head<DIR>(sp, U, g, hb, nbr, parity);
        staple_pipe_tail<DIR>(sp, U, g, hb, nbr, parity, NoHook());
        A = sp.stap;
    } else {
        A = staple<GROUP, DIR>(U, g, hb, nbr, parity);
    }
    mat<GROUP> Ul = load_link_buf(r_own, g, hb, (mat<GROUP>*)nullptr);
    overrelax_link(Ul, A, fm);
    store_link_buf(r_own, g, hb, Ul);
}

// ---------------------------------------------------------------------------------------------
// plaquette sums (SU(N)/suncpp/Measurements/Plq.h:46-94): per site the 3 space-space planes
// (xy, xz, yz) and the 3 space-time planes (xt, yt, zt); ReTr[U_mu(x) U_nu(x+mu) U_mu(x+nu)^+ U_nu(x)^+]

template <int GROUP>
__device__ __forceinline__ double retr_plaquette(const real2* __restrict__ U, const Geom& g, const Site& x,
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

// SU(3), the form the kernel uses: ReTr[(U_mu(x) U_nu(x+mu)) (U_nu(x) U_mu(x+nu))^+], both
// products SU(3) matrices, so two rows each come from products and the third from the cross
// product (whose conjugation drops out of the real part).  Buffer addressing as in the update.
__device__ __forceinline__ cplx cross_c(cplx a1, cplx a2, cplx b1, cplx b2)   // a1 b2 - a2 b1
{
    return csub(cmul(a1, b2), cmul(a2, b1));
}
// a = two rows of U_mu(x), d = two rows of U_nu(x) (kept by the caller: a site's four own links
// serve three planes each), b = U_nu(x+mu), c = U_mu(x+nu) loaded here.  ob = byte offset of the
// other parity's half of a direction's array plus the site's own offset (link_rsrc_both).
// the arithmetic: a, d two rows of U_mu(x), U_nu(x); rb = U_nu(x+mu), rc = U_mu(x+nu) as loaded
__device__ __forceinline__ double retr_plaquette_su3_from(const rows2& a, const rows2& d, const raw_link& rb, const raw_link& rc)
{
#ifdef QCD_PLAQ_ABL   // diagnostic build: loads kept, arithmetic of a plane reduced to a few additions
    double acc = 0.0;
#pragma unroll
    for (int c = 0; c < 6; c++) acc += (double)rb.c[c].x + (double)rc.c[c].y;
    return acc + (double)a.e[0][0].re + (double)d.e[1][1].im;
#endif
    const su3 b = as_su3(rb);
    const su3 c = as_su3(rc);
    cplx p[2][3], q[2][3];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            p[i][j] = dot3(a.e[i][0], a.e[i][1], a.e[i][2], b.e[0][j], b.e[1][j], b.e[2][j]);
            q[i][j] = dot3(d.e[i][0], d.e[i][1], d.e[i][2], c.e[0][j], c.e[1][j], c.e[2][j]);
        }
    double re = 0.0;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) re += p[i][j].re * q[i][j].re + p[i][j].im * q[i][j].im;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        const cplx cp = cross_c(p[0][j1], p[0][j2], p[1][j1], p[1][j2]);
        const cplx cq = cross_c(q[0][j1], q[0][j2], q[1][j1], q[1][j2]);
        re += cp.re * cq.re + cp.im * cq.im;
    }
    return re;
}
template <int AUX = 0>
__device__ __forceinline__ double retr_plaquette_su3(const real2* __restrict__ U, const Geom& g, const unsigned ob,
                                                     const NbrOff& n, const int mu, const int nu,
                                                     const rows2& a, const rows2& d)
{
    const raw_link rb = issue_link<AUX>(link_rsrc_both<3>(U, g, nu), g, (int)(ob + (unsigned)off_fwd(n, mu)));
    const raw_link rc = issue_link<AUX>(link_rsrc_both<3>(U, g, mu), g, (int)(ob + (unsigned)off_fwd(n, nu)));
    return retr_plaquette_su3_from(a, d, rb, rc);
}

// Thread -> site for the measurement kernels.  Both parities side by side (y fastest), and the
// workgroups dealt to the XCDs in eight contiguous runs, so that the requests for a link -- as a
// site's own link and as the neighbour of three others -- come from the same wave or from
// workgroups that run on one XCD at about the same time and find the line in that XCD's L2.  (One
// parity after the other, as the update pass has it, puts them 100 us apart: the plaquette kernel
// then fetched 3.4 times the lattice per call; plaquette 205 -> 158 us at 32^4.)
// False: no site for this thread.
template <int BLOCK>
__device__ __forceinline__ bool measurement_site(const Geom& g, Site& x)
{
    const int nb = gridDim.x;
#ifdef QCD_PLAQ_NO_XCD
    const int lb = blockIdx.x;
#else
    const int lb = (nb % 8 == 0) ? (blockIdx.x % 8) * (nb / 8) + blockIdx.x / 8 : blockIdx.x;
#endif
    int r = lb * BLOCK + threadIdx.x;
    if (r >= 2 * g.vh_int) return false;
    x.y = r % g.ny; r /= g.ny;
    x.z = r % g.nz; r /= g.nz;
    x.t = r % g.nt;
    x.xs = r / g.nt;
    return true;
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

// site_vals (may be null): the (space-like, time-like) sums of every site, index par * vh_int + interior
// half index, for the profiles below.
template <int GROUP, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_plaquette(const real2* __restrict__ U, const Geom g,
                                                     double2* __restrict__ partial, double2* __restrict__ site_vals)
{
    double spat = 0.0, temp = 0.0;
    Site x;
    if (measurement_site<BLOCK>(g, x)) {
        const int par = site_parity(g, x);
        if (GROUP == 3 && !g.odd) {
            const unsigned hb = (unsigned)half_index(g, x) * (unsigned)CB;
            const unsigned pbytes = (unsigned)ncomp<3>::value * (unsigned)g.vh_alloc * (unsigned)CB;
            const unsigned own_off = (par ? pbytes : 0u) + hb, other_off = (par ? 0u : pbytes) + hb;
            const NbrOff n = neighbour_offsets(g, x);
            rows2 own[4];
#pragma unroll
            for (int d = 0; d < 4; d++) own[d] = as_rows2(issue_link(link_rsrc_both<3>(U, g, d), g, (int)own_off));
            spat += retr_plaquette_su3(U, g, other_off, n, 0, 1, own[0], own[1]);
            spat += retr_plaquette_su3(U, g, other_off, n, 0, 2, own[0], own[2]);
            spat += retr_plaquette_su3(U, g, other_off, n, 1, 2, own[1], own[2]);
            temp += retr_plaquette_su3(U, g, other_off, n, 0, 3, own[0], own[3]);
            temp += retr_plaquette_su3(U, g, other_off, n, 1, 3, own[1], own[3]);
            temp += retr_plaquette_su3(U, g, other_off, n, 2, 3, own[2], own[3]);
        } else {
            spat += retr_plaquette<GROUP>(U, g, x, 0, 1);
            spat += retr_plaquette<GROUP>(U, g, x, 0, 2);
            spat += retr_plaquette<GROUP>(U, g, x, 1, 2);
            temp += retr_plaquette<GROUP>(U, g, x, 0, 3);
            temp += retr_plaquette<GROUP>(U, g, x, 1, 3);
            temp += retr_plaquette<GROUP>(U, g, x, 2, 3);
        }
        if (site_vals) {
            const int h = ((x.xs * g.nt + x.t) * g.nz + x.z) * g.nyh + (x.y >> 1);
            site_vals[(size_t)par * g.vh_int + h] = make_double2(spat, temp);
        }
    }
    block_reduce_store<BLOCK>(spat, temp, partial);
}

// The SU(3) plaquette sums with every thread marching XSEG sites along x.  In storage order x is
// the slowest index: the links U_y, U_z, U_t at x + xhat that the three planes containing x need
// are 128 workgroups away from the one that loads them as its own, which is two rounds of the
// resident workgroups -- three requests per site that never find their line in the L2.  A thread
// that goes on to x + xhat itself keeps them: they are requested once, serve the x planes of site x
// as neighbours and all planes of site x + xhat as its own links.  13 link requests per site instead
// of 16 (plus 3 per segment start), and the three dropped are the ones that always missed.
// Threads: (y, z, t) cross-section in storage order times Nx / XSEG segments; same partial /
// site_vals outputs.  (Workgroups shaped as tiles in (z, t) instead of rows in storage order, and
// workgroups of 512, were measured as well: 144 and 151 us per call against 141.)
// The kernel needs 257 registers in double precision, so a SIMD holds one wavefront and the 2048 of a 32^4 call
// run in two rounds; capped at 256 (two wavefronts per SIMD, all resident) it takes the same 140 us with 7.4 M
// L2 misses instead of 5.6 M: twice the columns in flight ask for twice the lines per step.
// Cache policy of the marching kernel's requests (2 = non-temporal, see staple_aux): the neighbours inside the slice
// were asked for by their owners one step earlier and are not needed after this step: with the hint on them a 32^4 call
// has 4.5 M L2 misses instead of 5.6 M and takes 132.8 us instead of 134.3 (back-to-back calls); on the links of the
// next slice or on U_x(x+nu) it costs time.
#ifndef QCD_PLAQ_NT_NXT
#define QCD_PLAQ_NT_NXT 0
#define QCD_PLAQ_NT_IN 2
#define QCD_PLAQ_NT_C 0
#endif
template <int BLOCK, int XSEG>
__global__ __launch_bounds__(BLOCK) void k_plaquette_march(const real2* __restrict__ U, const Geom g,
                                                           double2* __restrict__ partial, double2* __restrict__ site_vals)
{
    double spat = 0.0, temp = 0.0;
    const int nb = gridDim.x;
    const int lb = (nb % 8 == 0) ? (blockIdx.x % 8) * (nb / 8) + blockIdx.x / 8 : blockIdx.x;
    int r = lb * BLOCK + (int)threadIdx.x;
    const int seg = r / g.gslice;
    if (seg * XSEG < g.nx) {
        r -= seg * g.gslice;
        Site x;
        x.y = r % g.ny; r /= g.ny;
        x.z = r % g.nz;
        x.t = r / g.nz;
        x.xs = seg * XSEG;
        const unsigned pbytes = (unsigned)ncomp<3>::value * (unsigned)g.vh_alloc * (unsigned)CB;
        rows2 own[4];
        {
            const int par = site_parity(g, x);
            const unsigned own_off = (par ? pbytes : 0u) + (unsigned)half_index(g, x) * (unsigned)CB;
#pragma unroll
            for (int d = 0; d < 4; d++) own[d] = as_rows2(issue_link(link_rsrc_both<3>(U, g, d), g, (int)own_off));
        }
        const int steps = (g.nx - x.xs) < XSEG ? (g.nx - x.xs) : XSEG;
#pragma unroll 1
        for (int s = 0; s < steps; s++) {
            const int par = site_parity(g, x);
            const unsigned hb = (unsigned)half_index(g, x) * (unsigned)CB;
            const unsigned other_off = (par ? 0u : pbytes) + hb;
            const NbrOff n = neighbour_offsets(g, x);
            // the next site's y, z, t links (as this site's x neighbours), requested first
            raw_link nxt[3];
#pragma unroll
            for (int d = 1; d < 4; d++) nxt[d - 1] = issue_link<QCD_PLAQ_NT_NXT>(link_rsrc_both<3>(U, g, d), g, (int)(other_off + (unsigned)n.f0));
            double sp = 0.0, tp = 0.0;
            sp += retr_plaquette_su3<QCD_PLAQ_NT_IN>(U, g, other_off, n, 1, 2, own[1], own[2]);
            tp += retr_plaquette_su3<QCD_PLAQ_NT_IN>(U, g, other_off, n, 1, 3, own[1], own[3]);
            tp += retr_plaquette_su3<QCD_PLAQ_NT_IN>(U, g, other_off, n, 2, 3, own[2], own[3]);
            // planes with x: U_x(x) U_nu(x+xhat) (U_nu(x) U_x(x+nu))^+
            {
                const raw_link c1 = issue_link<QCD_PLAQ_NT_C>(link_rsrc_both<3>(U, g, 0), g, (int)(other_off + (unsigned)n.f1));
                sp += retr_plaquette_su3_from(own[0], own[1], nxt[0], c1);
                const raw_link c2 = issue_link<QCD_PLAQ_NT_C>(link_rsrc_both<3>(U, g, 0), g, (int)(other_off + (unsigned)n.f2));
                sp += retr_plaquette_su3_from(own[0], own[2], nxt[1], c2);
                const raw_link c3 = issue_link<QCD_PLAQ_NT_C>(link_rsrc_both<3>(U, g, 0), g, (int)(other_off + (unsigned)n.f3));
                tp += retr_plaquette_su3_from(own[0], own[3], nxt[2], c3);
            }
            if (site_vals) {
                const int h = ((x.xs * g.nt + x.t) * g.nz + x.z) * g.nyh + (x.y >> 1);
                site_vals[(size_t)par * g.vh_int + h] = make_double2(sp, tp);
            }
            spat += sp;
            temp += tp;
            if (s + 1 < steps) {
                // on to x + xhat: its y, z, t links are here, its x link is requested now
                x.xs++;
                const int par2 = par ^ 1;
                const unsigned own2 = (par2 ? pbytes : 0u) + (unsigned)half_index(g, x) * (unsigned)CB;
                own[0] = as_rows2(issue_link(link_rsrc_both<3>(U, g, 0), g, (int)own2));
#pragma unroll
                for (int d = 1; d < 4; d++) own[d] = as_rows2(nxt[d - 1]);
            }
        }
    }
    block_reduce_store<BLOCK>(spat, temp, partial);
}

// sums n double2 partials per output pair in a fixed order (strided, then a tree); workgroup o
// does output pair o: out[2*o], out[2*o+1].  (One workgroup looping over the outputs took 30 us for
// the 32 planes of a profile.  Folding this into the measurement kernels -- the workgroup that
// finishes last adds up -- was tried and is worse: the device-scope fences it needs write back and
// invalidate the XCDs' L2s, plaquette call 200 -> 390 us.)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_reduce(const double2* __restrict__ partial, const int n, double* __restrict__ out)
{
    __shared__ double sa[BLOCK], sb[BLOCK];
    const int o = blockIdx.x;
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
}

// ---------------------------------------------------------------------------------------------
// Wilson loop W(R,T) (kernel lattice_measurement_wilson, SU(N)/suncl/wilson_loop.cl:54-453): per
// site p and spatial direction i, ReTr[Tline(p) Rline_i(p + T t) Tline(p + R i)^+ Rline_i(p)^+].
// partial[block] = (sum over the block's sites and the three directions, 0)

template <int GROUP>
__device__ __forceinline__ mat<GROUP> link_line(const real2* __restrict__ U, const Geom& g, Site& q, int dir, int len)
{
    mat<GROUP> m = load_site_link<GROUP>(U, g, dir, q);
    for (int k = 1; k < len; k++) {
        q = fwd(g, q, dir);
        m = mul(m, load_site_link<GROUP>(U, g, dir, q));
    }
    q = fwd(g, q, dir);   // the site behind the line's end
    return m;
}

template <int GROUP, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_wilson(const real2* __restrict__ U, const Geom g, const int R, const int T,
                                                  double2* __restrict__ partial)
{
    double sum = 0.0;
    Site p;
    if (measurement_site<BLOCK>(g, p)) {
        Site pT = p;
        const mat<GROUP> m1 = link_line<GROUP>(U, g, pT, 3, T);
#pragma unroll 1
        for (int i = 0; i < 3; i++) {
            Site pR = p, q2 = pT;
            const mat<GROUP> m4 = link_line<GROUP>(U, g, pR, i, R);
            const mat<GROUP> m2 = link_line<GROUP>(U, g, q2, i, R);
            const mat<GROUP> m3 = link_line<GROUP>(U, g, pR, 3, T);
            sum += trace(mul(mul(mul(m1, m2), herm(m3)), herm(m4))).re;
        }
    }
    block_reduce_store<BLOCK>(sum, 0.0, partial);
}

// SU(3) on an uncut lattice with even extents: the same loop with every product kept as its first two rows (a line
// and the two halves of the loop are SU(3) matrices, so a third row is the cross product of the other two and is only
// formed where a matrix stands on the right of a product), buffer addressing, and the loop closed as
// ReTr[(Tline(p) Rline_i(p + T t)) (Rline_i(p) Tline(p + R i))^+] with the rule of retr_plaquette_su3_from.
__device__ __forceinline__ raw_link issue_site_link(const real2* __restrict__ U, const Geom& g, const int dir, const Site& s)
{
    const unsigned pbytes = (unsigned)ncomp<3>::value * (unsigned)g.vh_alloc * (unsigned)CB;
    const unsigned off = (site_parity(g, s) ? pbytes : 0u) + (unsigned)half_index(g, s) * (unsigned)CB;
    return issue_link(link_rsrc_both<3>(U, g, dir), g, (int)off);
}
__device__ __forceinline__ rows2 mul_rows(const rows2& a, const su3& b)
{
    rows2 p;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) p.e[i][j] = dot3(a.e[i][0], a.e[i][1], a.e[i][2], b.e[0][j], b.e[1][j], b.e[2][j]);
    return p;
}
__device__ __forceinline__ su3 completed(const rows2& a)
{
    su3 m;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) m.e[i][j] = a.e[i][j];
    reconstruct(m);
    return m;
}
__device__ __forceinline__ rows2 link_line_rows(const real2* __restrict__ U, const Geom& g, Site& q, const int dir, const int len)
{
    rows2 m = as_rows2(issue_site_link(U, g, dir, q));
    for (int k = 1; k < len; k++) {
        q = fwd(g, q, dir);
        m = mul_rows(m, as_su3(issue_site_link(U, g, dir, q)));
    }
    q = fwd(g, q, dir);   // the site behind the line's end
    return m;
}
// Re Tr(P Q^+) of two SU(3) matrices given by their first two rows
__device__ __forceinline__ double retr_rows(const rows2& p, const rows2& q)
{
    double re = 0.0;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) re += (double)(p.e[i][j].re * q.e[i][j].re + p.e[i][j].im * q.e[i][j].im);
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        const cplx cp = cross_c(p.e[0][j1], p.e[0][j2], p.e[1][j1], p.e[1][j2]);
        const cplx cq = cross_c(q.e[0][j1], q.e[0][j2], q.e[1][j1], q.e[1][j2]);
        re += (double)(cp.re * cq.re + cp.im * cq.im);
    }
    return re;
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_wilson_rows(const real2* __restrict__ U, const Geom g, const int R, const int T,
                                                       double2* __restrict__ partial)
{
    double sum = 0.0;
    Site p;
    if (measurement_site<BLOCK>(g, p)) {
        Site pT = p;
        const rows2 m1 = link_line_rows(U, g, pT, 3, T);
#pragma unroll 1
        for (int i = 0; i < 3; i++) {
            Site pR = p, q2 = pT;
            const rows2 m4 = link_line_rows(U, g, pR, i, R);
            const rows2 m2 = link_line_rows(U, g, q2, i, R);
            const rows2 m3 = link_line_rows(U, g, pR, 3, T);
            sum += retr_rows(mul_rows(m1, completed(m2)), mul_rows(m4, completed(m3)));
        }
    }
    block_reduce_store<BLOCK>(sum, 0.0, partial);
}

// ---------------------------------------------------------------------------------------------
// Polyakov loop (SU(N)/suncl/su3cpu.cpp:1383-1433; kernel lattice_polyakov,
// SU(N)/suncl/polyakov.cl:53-141): P = Tr prod_{t=0}^{Nt-1} U_T(x,t), left to right from t = 0.
// partial[0][block] = (sum Re P, sum Im P), partial[1][block] = (sum |P|^2, sum |P|^4);
// loops[y + Ny (z + Nz xs)] = P of that space site, for the profiles below.

template <int GROUP, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_polyakov(const real2* __restrict__ U, const Geom g,
                                                    double2* __restrict__ partial, const int nblocks,
                                                    double2* __restrict__ loops)
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
        // four links requested before the first of them is multiplied in: one thread per line is a chain of Nt dependent
        // products, and with one request in flight per thread the kernel waits a memory latency per link
        for (int t0 = 1; t0 < g.nt; t0 += 4) {
            mat<GROUP> l[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                x.t = (t0 + k < g.nt) ? t0 + k : g.nt - 1;
                l[k] = load_site_link<GROUP>(U, g, 3, x);
            }
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (t0 + k < g.nt) m = mul(m, l[k]);
        }
        const cplx tr = trace(m);
        re = tr.re;
        im = tr.im;
        p2 = re * re + im * im;
        p4 = p2 * p2;
        loops[t] = make_double2(re, im);
    }
    // two independent reductions; the helper owns its LDS, so run it twice through a barrier
    block_reduce_store<BLOCK>(re, im, partial);
    __syncthreads();
    block_reduce_store<BLOCK>(p2, p4, partial + nblocks);
}

// The same with a line cut into four segments: wavefront q of a workgroup multiplies the links t in [q * ceil(Nt/4), ...) of the
// workgroup's 64 space sites (coalesced as before, four times the threads and requests in flight: one thread per line leaves
// half the SIMDs of the device without a wavefront at 32^3 space sites), hands its partial product over through LDS, and wavefront 0
// multiplies the four in order.  P is the same product in another grouping: equal to the kernel above to rounding.
template <int GROUP, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_polyakov_split(const real2* __restrict__ U, const Geom g,
                                                          double2* __restrict__ partial, const int nblocks,
                                                          double2* __restrict__ loops)
{
    constexpr int SITES = BLOCK / 4;
    __shared__ real2 part[3][GROUP * GROUP][SITES];
    const int q = (int)threadIdx.x / SITES, ls = (int)threadIdx.x % SITES;
    const int t = blockIdx.x * SITES + ls;
    const int n3 = g.nx * g.ny * g.nz;
    const int len = (g.nt + 3) / 4;
    const int t0 = q * len, t1 = (t0 + len < g.nt) ? t0 + len : g.nt;
    mat<GROUP> m = unity<GROUP>();
    if (t < n3 && t0 < t1) {
        Site x;
        x.xs = t / (g.ny * g.nz);
        const int r = t - x.xs * (g.ny * g.nz);
        x.z = r / g.ny;
        x.y = r - x.z * g.ny;
        x.t = t0;
        m = load_site_link<GROUP>(U, g, 3, x);
        for (int tt = t0 + 1; tt < t1; tt += 4) {
            mat<GROUP> l[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                x.t = (tt + k < t1) ? tt + k : t1 - 1;
                l[k] = load_site_link<GROUP>(U, g, 3, x);
            }
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (tt + k < t1) m = mul(m, l[k]);
        }
    }
    if (q > 0) {
#pragma unroll
        for (int k = 0; k < GROUP * GROUP; k++) part[q - 1][k][ls] = make_real2(m.e[k / GROUP][k % GROUP].re, m.e[k / GROUP][k % GROUP].im);
    }
    __syncthreads();
    double re = 0.0, im = 0.0, p2 = 0.0, p4 = 0.0;
    if (q == 0 && t < n3) {
#pragma unroll 1
        for (int s = 0; s < 3; s++) {
            mat<GROUP> l;
#pragma unroll
            for (int k = 0; k < GROUP * GROUP; k++) {
                const real2 v = part[s][k][ls];
                l.e[k / GROUP][k % GROUP] = mk(v.x, v.y);
            }
            m = mul(m, l);
        }
        const cplx tr = trace(m);
        re = tr.re;
        im = tr.im;
        p2 = re * re + im * im;
        p4 = p2 * p2;
        loops[t] = make_double2(re, im);
    }
    block_reduce_store<BLOCK>(re, im, partial);
    __syncthreads();
    block_reduce_store<BLOCK>(p2, p4, partial + nblocks);
}

// ---------------------------------------------------------------------------------------------
// F_mu_nu traces (kernel lattice_measurement_plq with FMUNU / F0MU, SU(N)/suncl/sun_measurements_cl.cl:712-1175;
// lattice_retrace_plaquette3_F, SU(N)/suncl/su3_measurements_cl.cl:550-589; CPU twin
// SU(N)/suncl/su3cpu.cpp:1212-1381): per site and plane the plaquette matrix
// P = U_mu(x) U_nu(x+mu) U_mu(x+nu)^+ U_nu(x)^+ and  F_re = Im Tr(P lambda_a) / 2,
// F_im = -Re Tr(P lambda_a) / 2  for two Gell-Mann (SU(3)) or Pauli (SU(2)) matrices.  field 0: planes xy, xz, yz; field 1:
// planes xt, yt, zt.  partial[(3 * a + plane) * nblocks + block] = (sum F_re, sum F_im).

// Tr(P lambda_index), Gell-Mann matrices as su3cpu.cpp:775-920 writes them
__device__ __forceinline__ cplx trace_gell_mann(const su3& p, int index)
{
    switch (index) {
        case 1: return cadd(p.e[0][1], p.e[1][0]);
        case 2: { const cplx d = csub(p.e[0][1], p.e[1][0]); return mk(-d.im, d.re); }   // i (P12 - P21)
        case 3: return csub(p.e[0][0], p.e[1][1]);
        case 4: return cadd(p.e[0][2], p.e[2][0]);
        case 5: { const cplx d = csub(p.e[0][2], p.e[2][0]); return mk(-d.im, d.re); }
        case 6: return cadd(p.e[1][2], p.e[2][1]);
        case 7: { const cplx d = csub(p.e[1][2], p.e[2][1]); return mk(-d.im, d.re); }
        case 8: {
            const double a = 0.57735026918962576450914878050196, b = -1.1547005383792515290182975610039;
            return mk(p.e[0][0].re * a + p.e[1][1].re * a + p.e[2][2].re * b, p.e[0][0].im * a + p.e[1][1].im * a + p.e[2][2].im * b);
        }
        default: return trace(p);
    }
}

// Tr(P sigma_index), Pauli matrices (SU(N)/suncl/su2cpu.cpp:496-545; index 1..3)
__device__ __forceinline__ cplx trace_gell_mann(const su2& p, int index)
{
    switch (index) {
        case 1: return cadd(p.e[0][1], p.e[1][0]);
        case 2: { const cplx d = csub(p.e[0][1], p.e[1][0]); return mk(-d.im, d.re); }
        case 3: return csub(p.e[0][0], p.e[1][1]);
        default: return trace(p);
    }
}

template <int GROUP, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_fmunu(const real2* __restrict__ U, const Geom g, const int field, const int index1,
                                                 const int index2, double2* __restrict__ partial, const int nblocks)
{
    double f[12];
#pragma unroll
    for (int k = 0; k < 12; k++) f[k] = 0.0;
    Site x;
    if (measurement_site<BLOCK>(g, x)) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int mu = field ? k : (k == 2 ? 1 : 0);
            const int nu = field ? 3 : (k == 0 ? 1 : 2);
            const mat<GROUP> m1 = load_site_link<GROUP>(U, g, mu, x);
            const mat<GROUP> m2 = load_site_link<GROUP>(U, g, nu, fwd(g, x, mu));
            const mat<GROUP> m3 = herm(load_site_link<GROUP>(U, g, mu, fwd(g, x, nu)));
            const mat<GROUP> m4 = herm(load_site_link<GROUP>(U, g, nu, x));
            const mat<GROUP> p = mul(mul(mul(m1, m2), m3), m4);
            const cplx t1 = trace_gell_mann(p, index1), t2 = trace_gell_mann(p, index2);
            f[2 * k] = t1.im / 2;
            f[2 * k + 1] = -t1.re / 2;
            f[6 + 2 * k] = t2.im / 2;
            f[6 + 2 * k + 1] = -t2.re / 2;
        }
    }
#pragma unroll
    for (int k = 0; k < 6; k++) {
        block_reduce_store<BLOCK>(f[2 * k], f[2 * k + 1], partial + (size_t)k * nblocks);
        __syncthreads();
    }
}

// SU(3) on even extents, marching along x as k_plaquette_march does (same thread -> site map, the next slice's links of the
// planes with x carried over as the next site's own links), buffer addressing, a site's own links loaded once for its
// three planes.  P = (U_mu(x) U_nu(x+mu)) (U_nu(x) U_mu(x+nu))^+: two rows of each product from products, the third
// from the cross product, then the nine entries of P.  Link requests per site: 7 instead of 12 (field 0), 9 (field 1).
__device__ __forceinline__ void fmunu_plane_su3(const rows2& a, const rows2& d, const raw_link& rb, const raw_link& rc, const int index1,
                                                const int index2, double* f1, double* f2)
{
    const su3 b = as_su3(rb);
    const su3 c = as_su3(rc);
    cplx p[3][3], q[3][3];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) {
            p[i][j] = dot3(a.e[i][0], a.e[i][1], a.e[i][2], b.e[0][j], b.e[1][j], b.e[2][j]);
            q[i][j] = dot3(d.e[i][0], d.e[i][1], d.e[i][2], c.e[0][j], c.e[1][j], c.e[2][j]);
        }
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        p[2][j] = conj(cross_c(p[0][j1], p[0][j2], p[1][j1], p[1][j2]));
        q[2][j] = conj(cross_c(q[0][j1], q[0][j2], q[1][j1], q[1][j2]));
    }
    su3 w;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) w.e[i][j] = dot3c(p[i][0], p[i][1], p[i][2], q[j][0], q[j][1], q[j][2]);
    const cplx t1 = trace_gell_mann(w, index1), t2 = trace_gell_mann(w, index2);
    f1[0] += (double)t1.im / 2;
    f1[1] += -(double)t1.re / 2;
    f2[0] += (double)t2.im / 2;
    f2[1] += -(double)t2.re / 2;
}

template <int BLOCK, int XSEG, int FIELD>
__global__ __launch_bounds__(BLOCK) void k_fmunu_march(const real2* __restrict__ U, const Geom g, const int index1, const int index2,
                                                       double2* __restrict__ partial, const int nblocks)
{
    double f[12];
#pragma unroll
    for (int k = 0; k < 12; k++) f[k] = 0.0;
    const int nb = gridDim.x;
    const int lb = (nb % 8 == 0) ? (blockIdx.x % 8) * (nb / 8) + blockIdx.x / 8 : blockIdx.x;
    int r = lb * BLOCK + (int)threadIdx.x;
    const int seg = r / g.gslice;
    if (seg * XSEG < g.nx) {
        r -= seg * g.gslice;
        Site x;
        x.y = r % g.ny; r /= g.ny;
        x.z = r % g.nz;
        x.t = r / g.nz;
        x.xs = seg * XSEG;
        const unsigned pbytes = (unsigned)ncomp<3>::value * (unsigned)g.vh_alloc * (unsigned)CB;
        // the direction whose link at x + xhat is carried: field 0 carries y and z, field 1 carries t
        rows2 own[4];
        {
            const int par = site_parity(g, x);
            const unsigned own_off = (par ? pbytes : 0u) + (unsigned)half_index(g, x) * (unsigned)CB;
#pragma unroll
            for (int d = 0; d < 4; d++)
                if (FIELD || d < 3) own[d] = as_rows2(issue_link(link_rsrc_both<3>(U, g, d), g, (int)own_off));
        }
        const int steps = (g.nx - x.xs) < XSEG ? (g.nx - x.xs) : XSEG;
#pragma unroll 1
        for (int s = 0; s < steps; s++) {
            const int par = site_parity(g, x);
            const unsigned other_off = (par ? 0u : pbytes) + (unsigned)half_index(g, x) * (unsigned)CB;
            const NbrOff n = neighbour_offsets(g, x);
            const auto nbr = [&](int dir, int off) { return issue_link(link_rsrc_both<3>(U, g, dir), g, (int)(other_off + (unsigned)off)); };
            const auto nbr_last = [&](int dir, int off) { return issue_link<QCD_PLAQ_NT_IN>(link_rsrc_both<3>(U, g, dir), g, (int)(other_off + (unsigned)off)); };
            raw_link nxt[2];
            if (FIELD == 0) {
                nxt[0] = nbr(1, n.f0);
                nxt[1] = nbr(2, n.f0);
                {   // yz: k = 2
                    const raw_link b = nbr_last(2, n.f1), c = nbr_last(1, n.f2);
                    fmunu_plane_su3(own[1], own[2], b, c, index1, index2, f + 4, f + 10);
                }
                {   // xy: k = 0
                    const raw_link c = nbr(0, n.f1);
                    fmunu_plane_su3(own[0], own[1], nxt[0], c, index1, index2, f + 0, f + 6);
                }
                {   // xz: k = 1
                    const raw_link c = nbr(0, n.f2);
                    fmunu_plane_su3(own[0], own[2], nxt[1], c, index1, index2, f + 2, f + 8);
                }
            } else {
                nxt[0] = nbr(3, n.f0);
                {   // yt: k = 1
                    const raw_link b = nbr(3, n.f1), c = nbr(1, n.f3);
                    fmunu_plane_su3(own[1], own[3], b, c, index1, index2, f + 2, f + 8);
                }
                {   // zt: k = 2
                    const raw_link b = nbr(3, n.f2), c = nbr(2, n.f3);
                    fmunu_plane_su3(own[2], own[3], b, c, index1, index2, f + 4, f + 10);
                }
                {   // xt: k = 0
                    const raw_link c = nbr(0, n.f3);
                    fmunu_plane_su3(own[0], own[3], nxt[0], c, index1, index2, f + 0, f + 6);
                }
            }
            if (s + 1 < steps) {
                x.xs++;
                const unsigned own2 = ((par ^ 1) ? pbytes : 0u) + (unsigned)half_index(g, x) * (unsigned)CB;
                own[0] = as_rows2(issue_link(link_rsrc_both<3>(U, g, 0), g, (int)own2));
                if (FIELD == 0) {
                    own[1] = as_rows2(nxt[0]);
                    own[2] = as_rows2(nxt[1]);
                } else {
                    own[1] = as_rows2(issue_link(link_rsrc_both<3>(U, g, 1), g, (int)own2));
                    own[2] = as_rows2(issue_link(link_rsrc_both<3>(U, g, 2), g, (int)own2));
                    own[3] = as_rows2(nxt[0]);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 6; k++) {
        block_reduce_store<BLOCK>(f[2 * k], f[2 * k + 1], partial + (size_t)k * nblocks);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Profiles along a space axis (kernels lattice_action_diff_{x,y,z}, SU(N)/suncl/sun_measurements_cl.cl:186-616,
// and lattice_polyakov_diff_{x,y,z}, SU(N)/suncl/polyakov.cl:142-402): the plaquette sums and the
// Polyakov loops of the sites with coordinate b along `axis` (0 = x, 1 = y, 2 = z).  Workgroup
// (b, c) = blockIdx.x / chunks, blockIdx.x % chunks sums every chunks-th stretch of BLOCK sites of
// plane b; k_reduce adds the chunks of a plane in index order.  Both profiles read per-site values
// that the global measurement's kernel leaves behind (k_plaquette's site_vals, k_polyakov's loops),
// so a profile costs one pass over a small array, not another pass over the links.

// site number j of the plane: the coordinates other than `axis` run y, z, t, x from fastest to slowest
__device__ __forceinline__ Site plane_site(const Geom& g, int axis, int b, int j)
{
    int c[4];   // xs, y, z, t
    const int ext[4] = {g.nx, g.ny, g.nz, g.nt};
    const int order[4] = {1, 2, 3, 0};
    c[3] = 0;
    for (int k = 0; k < 4; k++) {
        const int d = order[k];
        if (d == axis) continue;
        c[d] = j % ext[d];
        j /= ext[d];
    }
    c[axis] = b;
    Site s;
    s.xs = c[0]; s.y = c[1]; s.z = c[2]; s.t = c[3];
    return s;
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_sites_profile(const double2* __restrict__ site_vals, const Geom g, const int axis,
                                                         const int chunks, const int plane_sites,
                                                         double2* __restrict__ partial)
{
    const int b = blockIdx.x / chunks, c = blockIdx.x - b * chunks;
    double spat = 0.0, temp = 0.0;
    for (int j = c * BLOCK + threadIdx.x; j < plane_sites; j += chunks * BLOCK) {
        const Site x = plane_site(g, axis, b, j);
        const int h = ((x.xs * g.nt + x.t) * g.nz + x.z) * g.nyh + (x.y >> 1);
        const double2 v = site_vals[(size_t)site_parity(g, x) * g.vh_int + h];
        spat += v.x;
        temp += v.y;
    }
    block_reduce_store<BLOCK>(spat, temp, partial);
}

// Polyakov profile from the per-site loops k_polyakov wrote: workgroup b adds the loops of the
// space sites with coordinate b along `axis`; partial[b] = (sum Re P, sum Im P).
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_loops_profile(const double2* __restrict__ loops, const Geom g, const int axis,
                                                         double2* __restrict__ partial)
{
    const int b = blockIdx.x;
    const int n3 = g.nx * g.ny * g.nz;
    const int plane_sites = n3 / (axis == 0 ? g.nx : (axis == 1 ? g.ny : g.nz));
    double re = 0.0, im = 0.0;
    for (int j = threadIdx.x; j < plane_sites; j += BLOCK) {
        int s;
        if (axis == 0) s = b * g.ny * g.nz + j;
        else if (axis == 1) s = b + g.ny * j;
        else {
            const int xs = j / g.ny;
            s = (j - xs * g.ny) + g.ny * (b + g.nz * xs);
        }
        const double2 p = loops[s];
        re += p.x;
        im += p.y;
    }
    block_reduce_store<BLOCK>(re, im, partial);
}

// ---------------------------------------------------------------------------------------------
// stand-alone re-unitarisation of every owned link (kernel lattice_GramSchmidt, suncl.cl:349-396)

template <int GROUP, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_reunit(real2* __restrict__ U, const Geom g)
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
__global__ __launch_bounds__(BLOCK) void k_init_cold(real2* __restrict__ U, const Geom g)
{
    const size_t n = (size_t)8 * ncomp<GROUP>::value * g.vh_alloc;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) {
        const int c = (int)((i / g.vh_alloc) % ncomp<GROUP>::value);
        const bool diag = (GROUP == 3) ? (c == 0 || c == 4) : (c == 0);
        U[i] = make_real2(diag ? (real)1 : (real)0, (real)0);
    }
}

// ---------------------------------------------------------------------------------------------
// hot start (kernels lattice_init_hot_{X,Y,Z,T}, SU(N)/suncl/suncl.cl:55-161): every link a
// random group element made from uniform numbers -- SU(3): product of three embedded SU(2)
// elements (lattice_random3, SU(N)/suncl/su3_update_cl.cl:164-225), SU(2): lattice_random2
// (SU(N)/suncl/su2_update_cl.cl:109-146).  The reference produces a fresh random buffer per
// direction (SU(N)/suncl/suncl.cpp:5236-5250); here stream h opens a fresh batch of 24 per
// direction, values 0..11 (SU(2): 0..3) for site 2h and 12..23 (SU(2): 4..7) for site 2h+1.

#define QCD_PI 3.1415926535897932384626433832795   // SU(N)/suncl/model.cl:120-123
#define QCD_PI2 6.2831853071795864769252867665590

__device__ __forceinline__ void hot_su2_angles(float r_alpha, float r_phi, float r_costh, double& a0, double& a1,
                                               double& a2, double& a3)
{
    const double alpha = QCD_PI * (double)r_alpha;
    const double phi = QCD_PI2 * (double)r_phi;
    const double costh = 2.0 * (double)r_costh - 1.0;
    const double sinal = sin(alpha);
    a0 = cos(alpha);
    const double sinph = sin(phi), cosph = cos(phi);
    const double sinth = sqrt(1.0 - costh * costh);
    const double t1 = sinal * sinth;
    a1 = t1 * cosph;
    a2 = t1 * sinph;
    a3 = sinal * costh;
}
// r: 12 values, r[k * stride]
__device__ __forceinline__ su3 hot_link(const float* r, int stride, su3*)
{
    double a0[3], a1[3], a2[3], a3[3];
#pragma unroll
    for (int k = 0; k < 3; k++) hot_su2_angles(r[k * stride], r[(4 + k) * stride], r[(8 + k) * stride], a0[k], a1[k], a2[k], a3[k]);
    constexpr int x = 0, y = 1, z = 2;
    su3 m;
    m.e[0][0] = mk(a0[x] * a0[y] - a3[x] * a3[y], a0[y] * a3[x] + a0[x] * a3[y]);
    m.e[0][1] = mk(-a0[x] * a1[y] * a1[z] + a0[z] * a2[x] - a0[x] * a2[y] * a2[z] - a1[z] * a2[y] * a3[x] + a1[y] * a2[z] * a3[x] - a1[x] * a3[z],
                   a0[z] * a1[x] + a0[x] * a1[z] * a2[y] - a0[x] * a1[y] * a2[z] - a1[y] * a1[z] * a3[x] - a2[y] * a2[z] * a3[x] + a2[x] * a3[z]);
    m.e[0][2] = mk(-a1[x] * a1[z] + a0[x] * a0[z] * a2[y] + a2[x] * a2[z] - a0[z] * a1[y] * a3[x] + a0[x] * a1[y] * a3[z] + a2[y] * a3[x] * a3[z],
                   a0[x] * a0[z] * a1[y] + a1[z] * a2[x] + a1[x] * a2[z] + a0[z] * a2[y] * a3[x] - a0[x] * a2[y] * a3[z] + a1[y] * a3[x] * a3[z]);
    m.e[1][0] = mk(-a0[y] * a2[x] - a1[x] * a3[y], a0[y] * a1[x] - a2[x] * a3[y]);
    m.e[1][1] = mk(a0[x] * a0[z] + a1[y] * a1[z] * a2[x] - a1[x] * a1[z] * a2[y] + a1[x] * a1[y] * a2[z] + a2[x] * a2[y] * a2[z] + a3[x] * a3[z],
                   -a1[x] * a1[y] * a1[z] - a1[z] * a2[x] * a2[y] + a1[y] * a2[x] * a2[z] - a1[x] * a2[y] * a2[z] - a0[z] * a3[x] + a0[x] * a3[z]);
    m.e[1][2] = mk(-a0[z] * a1[x] * a1[y] - a0[z] * a2[x] * a2[y] + a0[x] * a2[z] + a1[z] * a3[x] - a1[y] * a2[x] * a3[z] + a1[x] * a2[y] * a3[z],
                   a0[x] * a1[z] - a0[z] * a1[y] * a2[x] + a0[z] * a1[x] * a2[y] - a2[z] * a3[x] + a1[x] * a1[y] * a3[z] + a2[x] * a2[y] * a3[z]);
    reconstruct(m);
    return m;
}
// r: 4 values (the fourth is not used, as in the reference)
__device__ __forceinline__ su2 hot_link(const float* r, int stride, su2*)
{
    double a0, a1, a2, a3;
    hot_su2_angles(r[0], r[stride], r[2 * stride], a0, a1, a2, a3);
    su2 m;
    m.e[0][0] = mk(a0, a3);   // uv1 = (a0, a2, a3, a1) = (Re u1, Re u2, Im u1, Im u2)
    m.e[0][1] = mk(a2, a1);
    reconstruct(m);
    return m;
}

template <int GROUP, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_init_hot(real2* __restrict__ U, const Geom g, const RlxView rv,
                                                    unsigned long long* __restrict__ draws)
{
    __shared__ float rbuf[24 * BLOCK];
    constexpr int PER_LINK = (GROUP == 3) ? 12 : 4;
    const int t = blockIdx.x * BLOCK + threadIdx.x;   // stream of the owned slab
    unsigned long long used = 0;
    if (t < g.vh_int / g.share) {
        // the stream's site pairs: (xs, yh, z, tp + k tq), k = 0 .. share-1 (stream_of)
        int q = t;
        const int yh = q % g.nyh; q /= g.nyh;
        const int z = q % g.nz; q /= g.nz;
        const int tp = q % g.tq;
        const int xs = q / g.tq;
        RlxHome home;
        home.bind(rv, t);
        const int cur = rv.cur;
        RlxRegs r;
        home.load_state(r, cur);
        float* col = rbuf + threadIdx.x;
#pragma unroll 1
        for (int dir = 0; dir < 4; dir++)
#pragma unroll 1
            for (int k = 0; k < g.share; k++) {
                Site s0;                                   // the pair's site of parity 0
                s0.xs = xs; s0.z = z; s0.t = tp + k * g.tq;
                s0.y = 2 * yh + ((g.x0 + xs + z + s0.t) & 1);
                const int h = half_index(g, s0);
                const bool even_first = (s0.y & 1) == 0;   // site 2h is the one with even y
                rlx_batch(r, col, BLOCK);
                rlx_skip(r, home.q, home.rem);
                const mat<GROUP> a = hot_link(col, BLOCK, (mat<GROUP>*)nullptr);                     // site 2h
                const mat<GROUP> b = hot_link(col + PER_LINK * BLOCK, BLOCK, (mat<GROUP>*)nullptr);  // site 2h+1
                store_link(U, g, dir, 0, h, even_first ? a : b);
                store_link(U, g, dir, 1, h, even_first ? b : a);
                used += 2 * PER_LINK;
            }
        home.store_state(r, cur);   // in front of the stream's next batch, in the same slot: no pass has used it
    }
    used = block_sum_u64<BLOCK>(used);
    if ((threadIdx.x & 63) == 0 && used) atomicAdd(draws, used);
}

// ---------------------------------------------------------------------------------------------
// generator self-check (PRNG::check, SU(N)/random/random.cpp:339-402: device numbers against the
// CPU generator's, value by value): stream i delivers `batches` batches into out[(24 b + k) * ns + i]
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_rlx_dump(const unsigned* __restrict__ state, const int ns, const int q,
                                                    const int rem, const int batches, float* __restrict__ out)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= ns) return;
    RlxRegs r;
#pragma unroll
    for (int w = 0; w < 24; w++) r.s[w] = state[(size_t)w * ns + i];
    r.c = state[(size_t)24 * ns + i];
    for (int b = 0; b < batches; b++) {
        rlx_batch(r, out + (size_t)24 * b * ns + i, ns);
        rlx_skip(r, q, rem);
    }
}

// ---------------------------------------------------------------------------------------------
// self-check of fastmath.h: in[i] -> out[0][i] = log(1 - (float)in[i]) (the heat-bath's argument
// form), out[1][i] = sqrt(in[i]), out[2][i] = 1 / (in[i] + 1e-3), out[3][i] = 1/sqrt from sqrt_rcp
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_math_check(const double* __restrict__ in, const int n, double* __restrict__ out)
{
    __shared__ double2 logtab[LOG_TABLE_ENTRIES];
    MathFast::fill<BLOCK>(logtab);
    __syncthreads();
    MathFast fm;
    fm.tab = logtab;
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const double x = in[i];
    out[i] = fm.log_of(1.0 - (float)x);
    out[n + i] = fm.sqrt_of(x);
    out[2 * n + i] = fm.rcp(x + 1e-3);
    double s, is;
    fm.sqrt_rcp(x + 1e-3, s, is);
    out[3 * n + i] = is;
}

// self-check of MathFast::sincos_2pi: in[i] (a float) -> out[i] = cos(2 pi in[i]), out[n + i] = sin(2 pi in[i])
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_trig_check(const float* __restrict__ in, const int n, double* __restrict__ out)
{
    __shared__ double2 trigtab[TRIG_TABLE_ENTRIES];
    MathFast::fill_trig<BLOCK>(trigtab);
    __syncthreads();
    MathFast fm;
    fm.tab = nullptr;
    fm.trig = trigtab;
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    double s, c;
    fm.sincos_2pi(in[i], s, c);
    out[i] = c;
    out[n + i] = s;
}

// ---------------------------------------------------------------------------------------------
// reference device layout <-> engine layout.  `table` holds `nslices` x-slices starting at
// stored slice `slice0` (0 = left halo of a slab, or the first owned slice when halo = 0):
// rows of double4, element [lg + row*rowsize], lg = y + z*Ny + t*Ny*Nz + xl*Ny*Nz*Nt.

template <int GROUP, bool TO_ENGINE, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_convert(real2* __restrict__ U, const Geom g, real4* __restrict__ table,
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
            real4* q0 = table + (size_t)(0 * 4 + dir) * rowsize + lg;
            real4* q1 = table + (size_t)(1 * 4 + dir) * rowsize + lg;
            real4* q2 = table + (size_t)(2 * 4 + dir) * rowsize + lg;
            real2* p = U + comp_base<3>(g, dir, par) + h;
            const size_t st = g.vh_alloc;
            if (TO_ENGINE) {
                const real4 a = *q0, b = *q1, c = *q2;
                p[0 * st] = make_real2(a.x, b.x);  // u1
                p[1 * st] = make_real2(a.y, b.y);  // u2
                p[2 * st] = make_real2(a.z, b.z);  // u3
                p[3 * st] = make_real2(c.x, c.z);  // v1
                p[4 * st] = make_real2(c.y, c.w);  // v2
                p[5 * st] = make_real2(a.w, b.w);  // v3
            } else {
                const real2 u1 = p[0 * st], u2 = p[1 * st], u3 = p[2 * st];
                const real2 v1 = p[3 * st], v2 = p[4 * st], v3 = p[5 * st];
                *q0 = make_real4(u1.x, u2.x, u3.x, v3.x);
                *q1 = make_real4(u1.y, u2.y, u3.y, v3.y);
                *q2 = make_real4(v1.x, v2.x, v1.y, v2.y);
            }
        } else {
            real4* q0 = table + (size_t)dir * rowsize + lg;
            real2* p = U + comp_base<2>(g, dir, par) + h;
            const size_t st = g.vh_alloc;
            if (TO_ENGINE) {
                const real4 a = *q0;
                p[0 * st] = make_real2(a.x, a.z);  // u1
                p[1 * st] = make_real2(a.y, a.w);  // u2
            } else {
                const real2 u1 = p[0 * st], u2 = p[1 * st];
                *q0 = make_real4(u1.x, u2.x, u1.y, u2.y);
            }
        }
    }
}

// boundary slice <-> message buffer: buf[c*slice_h + r]
template <int GROUP, bool PACK, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_halo(real2* __restrict__ U, const Geom g, const int dir, const int par,
                                                const int stored_slice, real2* __restrict__ buf)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    constexpr int NC = ncomp<GROUP>::value;
    if (i >= NC * g.slice_h) return;
    const int c = i / g.slice_h;
    const int r = i - c * g.slice_h;
    real2* p = U + comp_base<GROUP>(g, dir, par) + (size_t)c * g.vh_alloc + (size_t)stored_slice * g.slice_h + r;
    if (PACK) buf[i] = *p;
    else *p = buf[i];
}

}  // namespace QCD_NS
