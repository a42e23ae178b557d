// Streaming bandwidth of this GPU for 16-byte-per-lane accesses (what the link arrays use):
// read-only, write-only and copy over a buffer larger than the 256 MB last-level cache.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench_bw.hip -o tools/bin/ubench_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_read(const double2* __restrict__ a, size_t n, double* out)
{
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double2 v = a[i];
        s += v.x + v.y;
    }
    if (s == 1.2345e300) *out = s;   // never true: keeps the loads
}
__global__ __launch_bounds__(256) void k_write(double2* __restrict__ a, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        a[i] = make_double2(1.0, 2.0);
}
__global__ __launch_bounds__(256) void k_copy(const double2* __restrict__ a, double2* __restrict__ b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}

int main(int argc, char** argv)
{
    const size_t mb = argc > 1 ? (size_t)atol(argv[1]) : 1024;
    const size_t n = mb * 1024 * 1024 / sizeof(double2);
    double2 *a, *b;
    double* out;
    CHECK(hipMalloc(&a, n * sizeof(double2)));
    CHECK(hipMalloc(&b, n * sizeof(double2)));
    CHECK(hipMalloc(&out, 8));
    CHECK(hipMemset(a, 0, n * sizeof(double2)));
    CHECK(hipMemset(b, 0, n * sizeof(double2)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int grids[] = {256 * 4, 256 * 8, 256 * 16, 256 * 32};
    for (int g : grids) {
        float best[3] = {1e9f, 1e9f, 1e9f};
        for (int rep = 0; rep < 5; rep++) {
            for (int k = 0; k < 3; k++) {
                CHECK(hipEventRecord(e0, 0));
                if (k == 0) hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, a, n, out);
                if (k == 1) hipLaunchKernelGGL(k_write, dim3(g), dim3(256), 0, 0, b, n);
                if (k == 2) hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, a, b, n);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best[k]) best[k] = ms;
            }
        }
        const double gb = n * sizeof(double2) / 1e9;
        printf("%4zu MB, grid %5d: read %6.2f TB/s  write %6.2f TB/s  copy %6.2f TB/s (read+write)\n", mb, g, gb / best[0], gb / best[1],
               2 * gb / best[2]);
    }
    return 0;
}
