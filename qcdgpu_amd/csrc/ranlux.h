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

namespace qcd {

constexpr int RLX_STATE_WORDS = 25;  // 24 words + carry
constexpr float RLX_TWOM24 = 0.000000059604644775390625f;

struct RlxRegs {
    unsigned s[24];
    unsigned c;
};

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

// skip nskip = 24*q + rem values
__host__ __device__ __forceinline__ void rlx_skip(RlxRegs& r, int q, int rem)
{
    for (int n = 0; n < q; n++) rlx_round(r);
    for (int n = 0; n < rem; n++) rlx_step_rotate(r);
}

// deliver one batch: 24 steps, value k written to out[k * stride]; includes the reference's
// fix-up of tiny values (random.cpp:566-570), done in fp32 exactly as there
template <typename OutT>
__host__ __device__ __forceinline__ void rlx_batch(RlxRegs& r, OutT* out, int stride)
{
#pragma unroll
    for (int k = 0; k < 24; k++) {
        const int i = 23 - k;
        const int j = (i + 10) % 24;
        QCD_RLX_STEP(r, i, j);
        const int jn = (i + 9) % 24;  // the reference's j24 after its decrement
        const unsigned ui = r.s[i];
        float uni = (float)ui * RLX_TWOM24;
        if (ui < 4096u) {
            uni = uni + RLX_TWOM24 * ((float)r.s[jn] * RLX_TWOM24);
            if (uni == 0.0f) uni = RLX_TWOM24 * RLX_TWOM24;
        }
        out[k * stride] = uni;
    }
}

}  // namespace qcd
