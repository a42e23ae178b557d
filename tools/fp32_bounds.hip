// Measures, over every 24-bit fraction r = k * 2^-24, the absolute error of the fp32 hardware
// transcendentals that the heat-bath uses to pre-screen its accept/reject test:
//   v_cos_f32(r)              against cos(2 pi r)
//   v_log_f32(1 - r) * ln 2   against log(1 - r)
// The margins in kernels.h (SCREEN_COS_ERR, SCREEN_LOG_ERR) must stay above these maxima.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out)
{
    __shared__ double sc[256], sl[256], sr[256];
    double ec = 0, el = 0, er = 0;
    for (unsigned kk = blockIdx.x * 256 + threadIdx.x; kk < (1u << 24); kk += gridDim.x * 256) {
        const float r = (float)kk * 0.000000059604644775390625f;
        const float c32 = __builtin_amdgcn_cosf(r);
        const double c64 = cospi(2.0 * (double)r);
        ec = fmax(ec, fabs((double)c32 - c64));
        if (kk) {
            const float l32 = __builtin_amdgcn_logf(1.0f - r) * 0.6931471805599453f;
            const double l64 = log(1.0 - (double)r);
            el = fmax(el, fabs((double)l32 - l64));
            er = fmax(er, fabs((double)l32 - l64) / fmax(fabs(l64), 1e-300));
        }
    }
    sc[threadIdx.x] = ec; sl[threadIdx.x] = el; sr[threadIdx.x] = er;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 256; i++) { ec = fmax(ec, sc[i]); el = fmax(el, sl[i]); er = fmax(er, sr[i]); }
        out[3 * blockIdx.x] = ec; out[3 * blockIdx.x + 1] = el; out[3 * blockIdx.x + 2] = er;
    }
}
int main()
{
    double* d; double h[3 * 1024];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double ec = 0, el = 0, er = 0;
    for (int i = 0; i < 1024; i++) { ec = fmax(ec, h[3*i]); el = fmax(el, h[3*i+1]); er = fmax(er, h[3*i+2]); }
    printf("max |v_cos_f32(r) - cos(2 pi r)| = %.3e\nmax |v_log_f32(1-r) ln2 - log(1-r)| = %.3e (relative %.3e)\n", ec, el, er);
    return 0;
}
