// ranlux.h -- per-stream RANLUX for the update kernels, written for registers.
//
// The generator is the reference's (SU(N)/random/random.cpp:504-574; device twin
// SU(N)/random/prngcl_ranlux.cl:203-563): a 24-word subtract-with-borrow sequence
// (lags 24 and 10) on 24-bit fractions, delivered in groups of 24 with `nskip` values thrown
// away in front of every group.  The reference keeps the state as floats k * 2^-24 and walks
// it with two run-time indices; a run-time index into a register array would send the state to
// scratch memory on gfx950, so this formulation differs in form (never in the numbers):
//   * words are the integers k (exact: every float the reference stores is such a multiple);
//   * the array is kept ROTATED so that the next step always works on word 23 and word 9;
//     a full round of 24 steps then uses only compile-time indices and leaves the rotation
//     unchanged; `nskip mod 24` single steps are each followed by a one-word rotation;
//   * one subtract-with-borrow (v_subb_co_u32) and one mask (v_and_b32) per step.
// A "batch" is one delivered group of 24.  The persistent state of a stream is the rotated
// state in front of a batch (24 words + carry) and the number of values of that batch already
// handed out.
#pragma once

#include <hip/hip_runtime.h>

// the C++ names of the library live in namespace QCD_NS: qcd, or qcd_f32 in the single-precision build, so that both
// libraries can be loaded into one process and a profile tells their kernels apart
#ifndef QCD_NS
#define QCD_NS qcd
#endif
namespace QCD_NS {

constexpr int RLX_STATE_WORDS = 25;  // 24 words + carry
constexpr float RLX_TWOM24 = 0.000000059604644775390625f;

struct RlxRegs {
    unsigned s[24];
    unsigned c;
};

// Device-memory form of a stream's state: the 24 words are 24-bit integers, so four of them fit
// three 32-bit words (the fourth one's bytes ride in the top bytes of the other three); 18 words
// + the carry = 19 instead of 25 per state, stored as five 16-byte quads per stream (the last word
// unused), quad q of stream i at word (q * ns + i) * 4: a wave moves a state with five 1 KiB
// accesses.  The state is read and written once per link update (an eighth of the update pass's
// memory traffic in unpacked form), the handful of byte moves is cheap beside that.
constexpr int RLX_PACKED_WORDS = 19;
constexpr int RLX_PACKED_QUADS = 5;
constexpr int RLX_SLOT_WORDS = 4 * RLX_PACKED_QUADS;   // per stream and slot in device memory
__host__ __device__ inline void rlx_pack4(const unsigned* s, unsigned& d0, unsigned& d1, unsigned& d2)
{
    d0 = s[0] | (s[3] << 24);
    d1 = s[1] | ((s[3] >> 8) << 24);
    d2 = s[2] | ((s[3] >> 16) << 24);
}
__host__ __device__ inline void rlx_unpack4(unsigned d0, unsigned d1, unsigned d2, unsigned* s)
{
    s[0] = d0 & 0xffffffu;
    s[1] = d1 & 0xffffffu;
    s[2] = d2 & 0xffffffu;
    s[3] = (d0 >> 24) | ((d1 >> 24) << 8) | ((d2 >> 24) << 16);
}
__host__ __device__ inline void rlx_pack(const RlxRegs& r, unsigned d[RLX_SLOT_WORDS])
{
    for (int j = 0; j < 6; j++) rlx_pack4(r.s + 4 * j, d[3 * j], d[3 * j + 1], d[3 * j + 2]);
    d[18] = r.c;
    d[19] = 0u;
}
__host__ __device__ inline void rlx_unpack(const unsigned d[RLX_SLOT_WORDS], RlxRegs& r)
{
    for (int j = 0; j < 6; j++) rlx_unpack4(d[3 * j], d[3 * j + 1], d[3 * j + 2], r.s + 4 * j);
    r.c = d[18];
}
// word k of stream i's packed state inside a slot of ns streams
inline size_t rlx_slot_index(int k, size_t ns, size_t i) { return ((size_t)(k / 4) * ns + i) * 4 + (size_t)(k % 4); }

// SU(N)/random/random.cpp:192-196
__host__ __device__ inline int rlx_nskip_for_level(int level)
{
    switch (level) {
        case 0: return 24;
        case 1: return 48;
        case 2: return 97;
        case 4: return 389;
        default: return 223;
    }
}

// one subtract-with-borrow step on compile-time positions (random.cpp:550-564)
#define QCD_RLX_STEP(r, I, J)                                      \
    do {                                                           \
        unsigned co_;                                              \
        unsigned t_ = __builtin_subc((r).s[J], (r).s[I], (r).c, &co_); \
        (r).c = co_;                                               \
        (r).s[I] = t_ & 0xFFFFFFu;                                 \
    } while (0)

// 24 steps, nothing delivered: the rotation is unchanged afterwards
__host__ __device__ __forceinline__ void rlx_round(RlxRegs& r)
{
#pragma unroll
    for (int k = 0; k < 24; k++) {
        const int i = 23 - k;
        const int j = (i + 10) % 24;
        QCD_RLX_STEP(r, i, j);
    }
}

// one step, then rotate by one word so that the next step is again on words 23 and 9
__host__ __device__ __forceinline__ void rlx_step_rotate(RlxRegs& r)
{
    QCD_RLX_STEP(r, 23, 9);
    const unsigned top = r.s[23];
#pragma unroll
    for (int k = 23; k > 0; k--) r.s[k] = r.s[k - 1];
    r.s[0] = top;
}

// skip nskip = 24*q + rem values (any luxury level)
__host__ __device__ __forceinline__ void rlx_skip(RlxRegs& r, int q, int rem)
{
    for (int n = 0; n < q; n++) rlx_round(r);
    for (int n = 0; n < rem; n++) rlx_step_rotate(r);
}

// the same for a skip count known at compile time: the closing rotation by REM words is a
// renaming of registers, not a copy
template <int NSKIP>
__host__ __device__ __forceinline__ void rlx_skip_static(RlxRegs& r)
{
    constexpr int Q = NSKIP / 24, REM = NSKIP % 24;
    for (int n = 0; n < Q; n++) rlx_round(r);
#pragma unroll
    for (int k = 0; k < REM; k++) {
        const int i = 23 - k;
        const int j = (i + 10) % 24;
        QCD_RLX_STEP(r, i, j);
    }
    if (REM) {
        RlxRegs t = r;
#pragma unroll
        for (int k = 0; k < 24; k++) r.s[k] = t.s[(k + 24 - REM) % 24];
    }
}

// deliver one batch: 24 steps, value k written to out[k * stride] as the float k' * 2^-24.
// The reference adds a correction to values below 2^-12 (random.cpp:566-570: the value plus
// 2^-24 times the word its j24 index points at after the step, in fp32); one value in 4096
// needs it, so the batch is produced without it and patched afterwards when its smallest value
// says so.  The word the correction reads is, for the first nine values of a batch, a word of
// the state in front of the batch (kept in `pre`), later one of the values already delivered.
template <typename OutT>
__host__ __device__ __forceinline__ void rlx_batch(RlxRegs& r, OutT* out, int stride)
{
    unsigned pre[9];
#pragma unroll
    for (int k = 0; k < 9; k++) pre[k] = r.s[k];
    unsigned lo = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < 24; k++) {
        const int i = 23 - k;
        const int j = (i + 10) % 24;
        QCD_RLX_STEP(r, i, j);
        const unsigned ui = r.s[i];
        lo = ui < lo ? ui : lo;
        out[k * stride] = (float)ui * RLX_TWOM24;
    }
    if (lo < 4096u) {
#pragma unroll
        for (int k = 0; k < 24; k++) {
            const unsigned ui = r.s[23 - k];
            if (ui < 4096u) {
                const unsigned sj = (k <= 8) ? pre[k <= 8 ? 8 - k : 0] : r.s[(32 - k) % 24];
                float uni = (float)ui * RLX_TWOM24;
                uni = uni + RLX_TWOM24 * ((float)sj * RLX_TWOM24);
                if (uni == 0.0f) uni = RLX_TWOM24 * RLX_TWOM24;
                out[k * stride] = uni;
            }
        }
    }
}

}  // namespace QCD_NS
