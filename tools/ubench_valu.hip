// Issue cost of the vector instructions the update kernel is made of (gfx950): dependent and
// independent chains of fp64 FMA, the RANLUX subtract-with-borrow step, fp32 FMA; one, two and
// four waves per SIMD.  Prints shader cycles per instruction per wave.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o tools/bin/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

constexpr int N = 8192;

template <int MODE>
__global__ void k(double* out, unsigned long long* cyc, double a, double b, unsigned ua)
{
    double x0 = threadIdx.x * 1e-3 + a, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    unsigned s0 = threadIdx.x + ua, s1 = s0 * 3, c = 0;
    float f0 = (float)x0;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < N / 32; i++) {
#pragma unroll
        for (int j = 0; j < 32; j++) {
            if (MODE == 0) x0 = __builtin_fma(x0, a, b);                       // dependent fp64 FMA
            if (MODE == 1) {                                                   // 4 independent chains
                if ((j & 3) == 0) x0 = __builtin_fma(x0, a, b);
                if ((j & 3) == 1) x1 = __builtin_fma(x1, a, b);
                if ((j & 3) == 2) x2 = __builtin_fma(x2, a, b);
                if ((j & 3) == 3) x3 = __builtin_fma(x3, a, b);
            }
            if (MODE == 2) f0 = __builtin_fmaf(f0, (float)a, (float)b);        // dependent fp32 FMA
            if (MODE == 3) {                                                   // RANLUX step pair
                unsigned co;
                unsigned t = __builtin_subc(s1, s0, c, &co);
                c = co;
                s0 = t & 0xFFFFFFu;
                unsigned t2 = __builtin_subc(s0, s1, c, &co);
                c = co;
                s1 = t2 & 0xFFFFFFu;
            }
            if (MODE == 4) x0 = x0 * a;                                        // dependent fp64 mul
            if (MODE == 5) x0 = x0 + a;                                        // dependent fp64 add
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + s0 + s1 + c + f0;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int waves_per_simd, int insts_per_iter)
{
    const int blocks = 256 * 4 * waves_per_simd;
    double* out;
    unsigned long long* cyc;
    hipMalloc(&out, (size_t)blocks * 64 * sizeof(double));
    hipMalloc(&cyc, blocks * sizeof(unsigned long long));
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, 0.999, 1e-3, 12345u);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-28s %d wave(s)/SIMD: median %.2f cyc/inst (p10 %.2f, p90 %.2f)\n", name, waves_per_simd,
           (double)h[blocks / 2] / (N * insts_per_iter), (double)h[blocks / 10] / (N * insts_per_iter),
           (double)h[blocks * 9 / 10] / (N * insts_per_iter));
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("fp64 fma dependent", w, 1);
        run<1>("fp64 fma 4 chains", w, 1);
        run<4>("fp64 mul dependent", w, 1);
        run<5>("fp64 add dependent", w, 1);
        run<2>("fp32 fma dependent", w, 1);
        run<3>("subb+and chain (per inst)", w, 4);
    }
    return 0;
}
