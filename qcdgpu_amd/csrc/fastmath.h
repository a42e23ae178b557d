// fastmath.h -- fp64 reciprocal, square root and logarithm for the heat-bath, built from the
// hardware seeds (v_rcp_f64, v_rsq_f64) and a 128-entry table instead of the device libm.
//
// Why: the heat-bath part of the update pass is issue-bound (every fp64 instruction on a wave's
// critical path costs its full 4-cycle slot), and the libm routines spend most of their
// instructions on cases that cannot occur here (denormals, infinities, arguments far from 1,
// double-double tails for the last half ulp).  Accuracy of these versions: <= 1-2 ulp on the
// argument ranges the heat-bath produces (checked against libm in tests); the CPU checker uses
// glibc, which the device libm does not match bit for bit either.
//
// Two interchangeable policies: MathLib (device libm, used off the hot path) and MathFast.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

// The precision the library is built for: links, their arithmetic and the heat-bath run in `real`
// (the reference's hgpu_float, SU(N)/kernel/complex.h:69-92, chosen there by -D PRECISION at kernel
// build time as well); sums over the lattice are always fp64.
#ifndef QCD_REAL
#define QCD_REAL double
#endif

// the C++ names of the library live in namespace QCD_NS: qcd, or qcd_f32 in the single-precision build, so that both
// libraries can be loaded into one process and a profile tells their kernels apart
#ifndef QCD_NS
#define QCD_NS qcd
#endif
namespace QCD_NS {

using real = QCD_REAL;
constexpr bool REAL_IS_DOUBLE = sizeof(real) == 8;
template <bool D> struct real2_of { using type = double2; };
template <> struct real2_of<false> { using type = float2; };
using real2 = real2_of<REAL_IS_DOUBLE>::type;
__host__ __device__ __forceinline__ real2 make_real2(real x, real y)
{
    real2 v;
    v.x = x;
    v.y = y;
    return v;
}
constexpr int CB = (int)sizeof(real2);   // bytes of one stored complex number
// a quad of the reference's device layout (hgpu_float4, SU(N)/suncl/model.cl:140-162)
template <bool D> struct real4_of { using type = double4; };
template <> struct real4_of<false> { using type = float4; };
using real4 = real4_of<REAL_IS_DOUBLE>::type;
__host__ __device__ __forceinline__ real4 make_real4(real x, real y, real z, real w)
{
    real4 v;
    v.x = x;
    v.y = y;
    v.z = z;
    v.w = w;
    return v;
}

struct MathLib {
    __device__ __forceinline__ float rcp(float x) const { return 1.0f / x; }
    __device__ __forceinline__ float sqrt_of(float x) const { return sqrtf(x); }
    __device__ __forceinline__ void sqrt_rcp(float x, float& s, float& is) const
    {
        s = sqrtf(x);
        is = 1.0f / s;
    }
    __device__ __forceinline__ float log_of(float x) const { return logf(x); }
    template <class T, class = typename std::enable_if<std::is_same<T, float>::value>::type>
    __device__ __forceinline__ float div(T a, T b) const { return a / b; }   // both float: mixed arguments take the fp64 form
    __device__ __forceinline__ double rcp(double x) const { return 1.0 / x; }
    __device__ __forceinline__ double sqrt_of(double x) const { return sqrt(x); }
    // s = sqrt(x), is = 1/s
    __device__ __forceinline__ void sqrt_rcp(double x, double& s, double& is) const
    {
        s = sqrt(x);
        is = 1.0 / s;
    }
    __device__ __forceinline__ double log_of(double x) const { return log(x); }
    __device__ __forceinline__ float rcp32(double x) const { return (float)(1.0 / x); }
    __device__ __forceinline__ double div(double a, double b) const { return a / b; }
    // cos(2 pi r), sin(2 pi r) for r in [0, 1): the reference forms 2*pi*r in fp64 and calls cos/sin;
    // cospi/sincospi reduce the argument exactly and are half the instructions
    __device__ __forceinline__ double cos_2pi(float r, double) const { return cospi(2.0 * (double)r); }
    __device__ __forceinline__ float cos_2pi(float r, float) const { return cospif(2.0f * r); }
    __device__ __forceinline__ void sincos_2pi(float r, double& s, double& c) const { sincospi(2.0 * (double)r, &s, &c); }
    __device__ __forceinline__ void sincos_2pi(float r, float& s, float& c) const { sincospif(2.0f * r, &s, &c); }
};

constexpr int LOG_TABLE_ENTRIES = 128;
constexpr int TRIG_TABLE_ENTRIES = 256;

struct MathFast {
    const double2* tab;   // LDS: (inv_j, -log(inv_j)), j = leading 7 fraction bits of the mantissa
    const double2* trig = nullptr;   // LDS: (cos, sin)(2 pi j / 256); needed by the fp64 cos_2pi / sincos_2pi only

    // one thread per entry fills the table (call with every thread of the block, then barrier)
    template <int BLOCK>
    __device__ static void fill_trig(double2* t)
    {
        for (int j = threadIdx.x; j < TRIG_TABLE_ENTRIES; j += BLOCK) {
            double sn, cs;
            sincospi((double)j * (2.0 / TRIG_TABLE_ENTRIES), &sn, &cs);
            t[j] = make_double2(cs, sn);
        }
    }
    // cos(2 pi r) and sin(2 pi r) for a RANLUX number r = k 2^-24: k = 2^16 hi + lo with |lo| <= 2^15, the angle of
    // hi from the table, the rest b = 2 pi lo 2^-24 (|b| <= pi/256) by its Taylor series, joined by the addition
    // theorems as  cos a + (cos a (cos b - 1) - sin a sin b).  Error <= 1.2e-16 absolute over all 2^24 arguments
    // (the reference's own cos(PI2 * r) with the rounded product is 6.4e-16 from the exact value); 25 instructions
    // for both against some 150 in the library routines, and a short dependent chain.
    __device__ __forceinline__ void sincos_2pi(float r, double& s, double& c) const
    {
#ifdef QCD_LIB_TRIG   // diagnostic build: the library routine, as before round 3
        sincospi(2.0 * (double)r, &s, &c);
        return;
#endif
        // (numbers below 2^-12 carry the generator's correction term, bits below 2^-24: r 2^24 is then no integer, and the
        // difference below is still exact in fp32 because hi = 0 there)
        const float rs = r * 16777216.0f;
        const int hi = ((int)rs + 0x8000) >> 16;
        const float lo = rs - (float)(hi << 16);
        const double2 t = trig[hi & (TRIG_TABLE_ENTRIES - 1)];
        const double b = (double)lo * 3.7450702829239286e-07;   // 2 pi 2^-24
        const double b2 = b * b;
        double cm = __builtin_fma(b2, -1.0 / 720.0, 1.0 / 24.0);
        cm = __builtin_fma(cm, b2, -0.5);
        cm = cm * b2;                                            // cos b - 1
        double sp = __builtin_fma(b2, -1.0 / 5040.0, 1.0 / 120.0);
        sp = __builtin_fma(sp, b2, -1.0 / 6.0);
        sp = sp * b2;
        const double sb = __builtin_fma(sp, b, b);               // sin b
        c = __builtin_fma(t.x, cm, -(t.y * sb)) + t.x;
        s = __builtin_fma(t.y, cm, t.x * sb) + t.y;
    }
    __device__ __forceinline__ double cos_2pi(float r, double) const
    {
#ifdef QCD_LIB_TRIG
        return cospi(2.0 * (double)r);
#endif
        double s, c;
        sincos_2pi(r, s, c);
        return c;
    }
    // single precision: the hardware's cosine and sine take their argument in revolutions (1.3e-7 absolute over all
    // 2^24 arguments, tools/fp32_bounds.hip)
#ifdef QCD_LIB_TRIG
    __device__ __forceinline__ float cos_2pi(float r, float) const { return cospif(2.0f * r); }
    __device__ __forceinline__ void sincos_2pi(float r, float& s, float& c) const { sincospif(2.0f * r, &s, &c); }
#else
    __device__ __forceinline__ float cos_2pi(float r, float) const { return __builtin_amdgcn_cosf(r); }
    __device__ __forceinline__ void sincos_2pi(float r, float& s, float& c) const
    {
        s = __builtin_amdgcn_sinf(r);
        c = __builtin_amdgcn_cosf(r);
    }
#endif

    // one thread per entry fills the table (call with every thread of the block, then barrier)
    template <int BLOCK>
    __device__ static void fill(double2* t)
    {
        for (int j = threadIdx.x; j < LOG_TABLE_ENTRIES; j += BLOCK) {
            // centre of the j-th mantissa interval of f in [0.5, 1); the last interval is
            // anchored at 1 so that arguments just below 1 lose nothing to cancellation
            const double fc = (j == LOG_TABLE_ENTRIES - 1) ? 1.0 : 0.5 * (1.0 + (j + 0.5) / LOG_TABLE_ENTRIES);
            const double inv = 1.0 / fc;
            t[j] = make_double2(inv, -log(inv));
        }
    }

    // single precision: the hardware's reciprocal, reciprocal square root and logarithm (1 ulp) with one
    // correction step where a product of the results is formed
    __device__ __forceinline__ float rcp(float x) const
    {
        const float y = __builtin_amdgcn_rcpf(x);
        return __builtin_fmaf(y, __builtin_fmaf(-x, y, 1.0f), y);
    }
    template <class T, class = typename std::enable_if<std::is_same<T, float>::value>::type>
    __device__ __forceinline__ float div(T a, T b) const { return a * rcp(b); }
    __device__ __forceinline__ void sqrt_rcp(float x, float& s, float& is) const
    {
        const float y = __builtin_amdgcn_rsqf(x);
        float g = x * y;
        const float h = 0.5f * y;
        g = __builtin_fmaf(__builtin_fmaf(-g, g, x), h, g);
        s = g;
        is = rcp(g);
    }
    __device__ __forceinline__ float sqrt_of(float x) const
    {
        const float y = __builtin_amdgcn_rsqf(x);
        float g = x * y;
        g = __builtin_fmaf(__builtin_fmaf(-g, g, x), 0.5f * y, g);
        return (x == 0.0f) ? 0.0f : g;
    }
    __device__ __forceinline__ float log_of(float x) const { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }

    // normal, positive x well inside the exponent range (here: norms of 2x2 blocks, 1e-3 .. 1e2)
    __device__ __forceinline__ double rcp(double x) const
    {
        double y = __builtin_amdgcn_rcp(x);
        double e = __builtin_fma(-x, y, 1.0);
        y = __builtin_fma(y, e, y);
        e = __builtin_fma(-x, y, 1.0);
        return __builtin_fma(y, e, y);
    }
    __device__ __forceinline__ float rcp32(double x) const { return __builtin_amdgcn_rcpf((float)x); }
    // a / b to about one ulp (b as for rcp)
    __device__ __forceinline__ double div(double a, double b) const { return a * rcp(b); }

    // x >= 0, not denormal; Goldschmidt iteration from the hardware reciprocal square root
    __device__ __forceinline__ void sqrt_rcp(double x, double& s, double& is) const
    {
        const double y = __builtin_amdgcn_rsq(x);
        double g = x * y;            // ~ sqrt(x)
        double h = 0.5 * y;          // ~ 1 / (2 sqrt(x))
        double r = __builtin_fma(-h, g, 0.5);
        g = __builtin_fma(g, r, g);
        h = __builtin_fma(h, r, h);
        r = __builtin_fma(-h, g, 0.5);
        h = __builtin_fma(h, r, h);
        const double d = __builtin_fma(-g, g, x);
        g = __builtin_fma(d, h, g);
        s = g;
        is = h + h;
    }
    __device__ __forceinline__ double sqrt_of(double x) const
    {
        const double y = __builtin_amdgcn_rsq(x);
        double g = x * y;
        double h = 0.5 * y;
        const double r = __builtin_fma(-h, g, 0.5);
        g = __builtin_fma(g, r, g);
        h = __builtin_fma(h, r, h);
        const double d = __builtin_fma(-g, g, x);
        g = __builtin_fma(d, h, g);
        return (x == 0.0) ? 0.0 : g;   // rsq(0) = inf
    }

    // x > 0, normal: x = f 2^e, f in [0.5, 1); f = (1 + z) / inv_j with |z| <= 2^-7, so
    // log x = e ln2 - log(inv_j) + log1p(z) holds exactly for the ROUNDED table value inv_j.
    // Relative error <= 1.4 ulp for x <= 1; for x just above 1 the two leading terms cancel and
    // the error is absolute, about 1e-16.
    __device__ __forceinline__ double log_of(double x) const
    {
        const double f = __builtin_amdgcn_frexp_mant(x);
        const int e = __builtin_amdgcn_frexp_exp(x);
        const int j = (__double2hiint(f) >> 13) & (LOG_TABLE_ENTRIES - 1);
        const double2 t = tab[j];
        const double z = __builtin_fma(f, t.x, -1.0);
        double p = __builtin_fma(z, 1.0 / 7.0, -1.0 / 6.0);
        p = __builtin_fma(p, z, 1.0 / 5.0);
        p = __builtin_fma(p, z, -1.0 / 4.0);
        p = __builtin_fma(p, z, 1.0 / 3.0);
        p = __builtin_fma(p, z, -0.5);
        p = p * z;
        p = __builtin_fma(p, z, z);   // log1p(z)
        return __builtin_fma((double)e, 0.693147180559945309417232121458, t.y) + p;
    }
};

}  // namespace QCD_NS
