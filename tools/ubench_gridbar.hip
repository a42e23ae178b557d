// What a grid-wide barrier inside one launch costs on the 8 XCDs of this GPU, with the data written before it visible
// to every workgroup after it: the price of running several update passes in one launch (generator state kept on chip).
// 256 workgroups of 512 threads with 120 KB of LDS each (one per CU, as the pair kernel runs), cooperative launch.
// Round r: workgroup b writes `per` quads of its own region (value r * 1000003 + index), barrier, reads the region of
// workgroup (b + 37) % n (another XCD) and counts wrong values.  Variants: stores plain / non-temporal / write-through;
// the barrier's release with and without the L2 write-back, its acquire with and without the L2 invalidate.
// Every spin loop gives up after a bounded number of polls and raises a flag, so the grid always drains.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench_gridbar.hip -o tools/bin/ubench_gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Args {
    u32x4* data;
    unsigned* counter;   // [0] arrivals, [1] give-up flag
    unsigned long long* bad;
    int per;             // quads per workgroup and round
    int rounds;
    int mode;            // stores: 0 plain, 1 non-temporal, 2 write-through (sc0 sc1)
    int fence;           // 0: agent-scope release + acquire fences, 1: none (counter only: how much is the fences)
};

template <int MODE>
__device__ __forceinline__ void put(u32x4* p, u32x4 v)
{
    if (MODE == 1) __builtin_nontemporal_store(v, p);
    else if (MODE == 2) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 16, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(v, r, 0, 0, 1 | 16);
    } else *p = v;
}

template <int MODE>
__global__ __launch_bounds__(512, 1) void k_rounds(const Args a)
{
    extern __shared__ unsigned lds[];
    const int n = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
    unsigned long long wrong = 0;
    if (tid == 0) lds[0] = 0;
    for (int r = 1; r <= a.rounds; r++) {
        u32x4* mine = a.data + (size_t)b * a.per;
        for (int i = tid; i < a.per; i += 512) {
            const unsigned v = (unsigned)r * 1000003u + (unsigned)i;
            u32x4 q;
            q.x = v; q.y = v ^ 0x55u; q.z = (unsigned)b; q.w = (unsigned)r;
            put<MODE>(mine + i, q);
        }
        // ---- barrier
        __syncthreads();
        if (tid == 0) {
            if (!a.fence) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(&a.counter[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)r * (unsigned)n;
            unsigned polls = 0;
            while (__hip_atomic_load(&a.counter[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(2);
                if (++polls > 4000000u || __hip_atomic_load(&a.counter[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(&a.counter[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    lds[0] = 1;
                    break;
                }
            }
            if (!a.fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (lds[0]) break;
        // ---- read what another workgroup wrote in this round
        const int o = (b + 37) % n;
        const u32x4* theirs = a.data + (size_t)o * a.per;
        for (int i = tid; i < a.per; i += 512) {
            const u32x4 q = theirs[i];
            const unsigned v = (unsigned)r * 1000003u + (unsigned)i;
            if (q.x != v || q.y != (v ^ 0x55u) || q.z != (unsigned)o || q.w != (unsigned)r) wrong++;
        }
        // nobody may overwrite a region before its reader is done with it: a second barrier (counter only)
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(&a.counter[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)r * (unsigned)n;
            unsigned polls = 0;
            while (__hip_atomic_load(&a.counter[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(2);
                if (++polls > 4000000u || __hip_atomic_load(&a.counter[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(&a.counter[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    lds[0] = 1;
                    break;
                }
            }
        }
        __syncthreads();
        if (lds[0]) break;
    }
    if (wrong) atomicAdd(a.bad, wrong);
}

int main()
{
    const int n = 256, rounds = 40;
    Args a;
    const int per_max = 16384;   // 256 KB per workgroup and round: 64 MB per round in all
    CHECK(hipMalloc(&a.data, (size_t)n * per_max * sizeof(u32x4)));
    CHECK(hipMalloc(&a.counter, 64));
    CHECK(hipMalloc(&a.bad, 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const size_t lds_bytes = 120 * 1024;
    CHECK(hipFuncSetAttribute((const void*)k_rounds<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    CHECK(hipFuncSetAttribute((const void*)k_rounds<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    CHECK(hipFuncSetAttribute((const void*)k_rounds<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    const char* names[3] = {"plain", "non-temporal", "write-through"};
    for (int fence = 0; fence < 2; fence++)
        for (int mode = 0; mode < 3; mode++)
            for (int per : {0, 1024, 16384}) {
                a.per = per;
                a.rounds = rounds;
                a.mode = mode;
                a.fence = fence;
                CHECK(hipMemset(a.counter, 0, 64));
                CHECK(hipMemset(a.bad, 0, 8));
                void* params[] = {(void*)&a};
                const void* fn = mode == 0 ? (const void*)k_rounds<0> : mode == 1 ? (const void*)k_rounds<1> : (const void*)k_rounds<2>;
                CHECK(hipEventRecord(e0, 0));
                CHECK(hipLaunchCooperativeKernel(fn, dim3(n), dim3(512), params, lds_bytes, 0));
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                unsigned c[4];
                unsigned long long bad;
                CHECK(hipMemcpy(c, a.counter, 16, hipMemcpyDeviceToHost));
                CHECK(hipMemcpy(&bad, a.bad, 8, hipMemcpyDeviceToHost));
                printf("%-14s stores, %s, %6.2f MB per round: %7.2f us per round (two barriers)  wrong values %llu%s\n", names[mode],
                       fence ? "no fences      " : "release+acquire", (double)n * per * 16 / 1e6, 1e3 * ms / rounds, bad, c[1] ? "  GAVE UP" : "");
                fflush(stdout);
                if (c[1]) return 2;
            }
    return 0;
}
