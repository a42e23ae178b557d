// What a re-read that misses the 4 MiB L2 of an XCD costs when the line is still in the 256 MiB
// memory-side cache: (1) read bandwidth over buffers from 32 MB to 1 GB read again and again,
// (2) a 1 GB stream in which every 16 bytes are read a second time `lag` megabytes later.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench_mall.hip -o tools/bin/ubench_mall
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_read(const double2* __restrict__ a, size_t n, double* out)
{
    double s = 0.0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const double2 v0 = a[i], v1 = a[i + stride], v2 = a[i + 2 * stride], v3 = a[i + 3 * stride];
        s += v0.x + v1.y + v2.x + v3.y;
    }
    for (; i < n; i += stride) s += a[i].x;
    if (s == 1.2345e300) *out = s;
}
// every element read twice: at its own turn and again `lag` elements later in the stream
__global__ __launch_bounds__(256) void k_reread(const double2* __restrict__ a, size_t n, size_t lag, double* out)
{
    double s = 0.0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + stride < n; i += 2 * stride) {
        const double2 v0 = a[i], v1 = a[i + stride];
        const size_t j0 = i >= lag ? i - lag : i, j1 = i + stride >= lag ? i + stride - lag : i + stride;
        const double2 w0 = a[j0], w1 = a[j1];
        s += v0.x + v1.y + w0.x + w1.y;
    }
    if (s == 1.2345e300) *out = s;
}

int main(int argc, char** argv)
{
    const size_t maxmb = 1024;
    const size_t nmax = maxmb * 1024 * 1024 / sizeof(double2);
    double2* a;
    double* out;
    CHECK(hipMalloc(&a, nmax * sizeof(double2)));
    CHECK(hipMalloc(&out, 8));
    CHECK(hipMemset(a, 0, nmax * sizeof(double2)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int g = 256 * 8;
    const size_t sizes[] = {16, 32, 64, 128, 192, 256, 384, 512, 1024};
    for (size_t mb : sizes) {
        const size_t n = mb * 1024 * 1024 / sizeof(double2);
        const int inner = (int)(2048 / mb) > 0 ? (int)(2048 / mb) : 1;
        float best = 1e9f;
        for (int rep = 0; rep < 4; rep++) {
            CHECK(hipEventRecord(e0, 0));
            for (int k = 0; k < inner; k++) hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, a, n, out);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms / inner < best) best = ms / inner;
        }
        printf("read %5zu MB x%d back to back: %7.1f us per pass, %6.2f TB/s\n", mb, inner, best * 1e3, mb * 1.048576e-3 / best);
    }
    const size_t lags_mb[] = {0, 1, 4, 16, 48, 96, 160, 400};
    for (size_t lmb : lags_mb) {
        const size_t lag = lmb * 1024 * 1024 / sizeof(double2);
        float best = 1e9f;
        for (int rep = 0; rep < 4; rep++) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_reread, dim3(g), dim3(256), 0, 0, a, nmax, lag, out);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("1 GB stream, every element read again %4zu MB later: %7.1f us, requested %6.2f TB/s, unique %6.2f TB/s\n", lmb, best * 1e3,
               2 * 1.073741824 / best, 1.073741824 / best);
    }
    return 0;
}
