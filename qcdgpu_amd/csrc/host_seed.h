// host_seed.h -- host-side seeding of the RANLUX streams and the closed-form "gid" tables.
//
// The reference derives every seed from the C library: srand(RANDSERIES), then rand()
// (SU(N)/random/random.cpp:174-182, 508); its device generator gives instance 0 the CPU
// generator's seed and every further instance the next rand() value (random.cpp:462-465).
// Stream i of this engine is seeded like instance i there.  rand() is restated here (glibc's
// TYPE_3 additive-feedback generator, degree 31, separation 3, 310 values discarded after
// seeding) so that the streams do not depend on the C library the host happens to run.
#pragma once

#include <cstdint>
#include <vector>

#include "ranlux.h"

// the C++ names of the library live in namespace QCD_NS: qcd, or qcd_f32 in the single-precision build, so that both
// libraries can be loaded into one process and a profile tells their kernels apart
#ifndef QCD_NS
#define QCD_NS qcd
#endif
namespace QCD_NS {

class GlibcRand {
public:
    explicit GlibcRand(unsigned seed)
    {
        if (seed == 0) seed = 1;
        int32_t word = (int32_t)seed;
        ring_[0] = (uint32_t)word;
        for (int i = 1; i < 31; i++) {
            const long hi = word / 127773, lo = word % 127773;
            long w = 16807 * lo - 2836 * hi;
            if (w < 0) w += 2147483647;
            word = (int32_t)w;
            ring_[i] = (uint32_t)word;
        }
        f_ = 3;
        b_ = 0;
        for (int i = 0; i < 310; i++) (void)next();
    }
    int next()
    {
        ring_[f_] += ring_[b_];
        const uint32_t r = ring_[f_] >> 1;
        f_ = (f_ + 1) % 31;
        b_ = (b_ + 1) % 31;
        return (int)r;
    }

private:
    uint32_t ring_[31];
    int f_, b_;
};

// RL_initialize_CPU (random.cpp:504-522) followed by the nskip values the reference discards
// in front of its first delivered group (RL_in24 starts at 0, random.cpp:69, 527-548); the
// result is the rotated state in front of batch 0.
inline void rlx_seed_state(int jseed, int nskip, RlxRegs& r)
{
    int js = jseed;
    for (int i = 0; i < 24; i++) {
        const int k = js / 53668;
        js = 40014 * (js - k * 53668) - k * 12211;
        if (js < 0) js += 2147483563;
        r.s[i] = (unsigned)(js % 16777216);
    }
    r.c = (r.s[23] == 0) ? 1u : 0u;
    rlx_skip(r, nskip / 24, nskip % 24);
}

// seeds of streams [first, first + count): stream 0 = rand() % 2147483647, stream i = the
// (i+1)-th rand() value
inline std::vector<int> stream_seeds(unsigned rand_series, long first, long count)
{
    GlibcRand g(rand_series);
    std::vector<int> out((size_t)count);
    int v = g.next();
    long idx = 0;
    if (first == 0 && count > 0) out[0] = v % 2147483647;
    while (idx < first + count - 1) {
        v = g.next();
        idx++;
        if (idx >= first) out[(size_t)(idx - first)] = v;
    }
    return out;
}

}  // namespace QCD_NS
