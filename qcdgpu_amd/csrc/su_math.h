// su_math.h -- register-resident SU(2)/SU(3) algebra for the gfx950 kernels.
//
// Every matrix lives in VGPRs as fully unrolled, statically indexed arrays of complex numbers
// of the library's precision (`real`, fastmath.h); there is no MFMA here on purpose (3x3 contractions are far below one MFMA tile).
// The term order inside each product follows the reference's sequential CPU path
// (SU(N)/suncpp/su3/algebra_su3.cpp:118-143, su2/algebra_su2.cpp:90-102) so that, apart from
// fused multiply-adds, the device rounds like the CPU checker does.
#pragma once

#include <hip/hip_runtime.h>

#include "fastmath.h"

// the C++ names of the library live in namespace QCD_NS: qcd, or qcd_f32 in the single-precision build, so that both
// libraries can be loaded into one process and a profile tells their kernels apart
#ifndef QCD_NS
#define QCD_NS qcd
#endif
namespace QCD_NS {

struct cplx {
    real re, im;
};

__device__ __forceinline__ cplx mk(real re, real im)
{
    cplx c;
    c.re = re;
    c.im = im;
    return c;
}
__device__ __forceinline__ cplx conj(cplx a) { return mk(a.re, -a.im); }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return mk(a.re + b.re, a.im + b.im); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return mk(a.re - b.re, a.im - b.im); }
// Single precision: complex products as packed pairs.  c += a * b is two v_pk_fma_f32 on (re, im) register
// pairs -- (a.re, a.re) * (b.re, b.im) + c, then (-a.im, a.im) * (b.im, b.re) + that -- against four scalar
// multiply-adds; the instruction selector folds the swaps into the operand selects of the packed form.  The
// float build is bound by instruction issue, not by memory (fastmath.h has the precision switch).
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk(cplx a) { return (v2f){(float)a.re, (float)a.im}; }
__device__ __forceinline__ cplx unpk(v2f a) { return mk((real)a.x, (real)a.y); }
__device__ __forceinline__ v2f pk_cmac(v2f c, v2f a, v2f b)
{
    c = __builtin_elementwise_fma(a.xx, b, c);
    return __builtin_elementwise_fma((v2f){-a.y, a.y}, b.yx, c);
}
__device__ __forceinline__ v2f pk_cmul(v2f a, v2f b)
{
    return __builtin_elementwise_fma((v2f){-a.y, a.y}, b.yx, a.xx * b);
}

// SU(N)/kernel/complex.h:371-376
__device__ __forceinline__ cplx cmul(cplx a, cplx b)
{
    if constexpr (!REAL_IS_DOUBLE) return unpk(pk_cmul(pk(a), pk(b)));
    return mk(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re);
}

template <int N>
struct mat {
    cplx e[N][N];  // e[row][col]
};
using su3 = mat<3>;
using su2 = mat<2>;

// one entry of a product: sum_k a_k * b_k, in the reference's term order
__device__ __forceinline__ cplx dot3(cplx a1, cplx a2, cplx a3, cplx b1, cplx b2, cplx b3)
{
    if constexpr (!REAL_IS_DOUBLE) return unpk(pk_cmac(pk_cmac(pk_cmul(pk(a1), pk(b1)), pk(a2), pk(b2)), pk(a3), pk(b3)));
    cplx c;
    c.re = -a1.im * b1.im + a1.re * b1.re - a2.im * b2.im + a2.re * b2.re - a3.im * b3.im + a3.re * b3.re;
    c.im = a1.re * b1.im + a1.im * b1.re + a2.re * b2.im + a2.im * b2.re + a3.re * b3.im + a3.im * b3.re;
    return c;
}
__device__ __forceinline__ cplx dot2(cplx a1, cplx a2, cplx b1, cplx b2)
{
    if constexpr (!REAL_IS_DOUBLE) return unpk(pk_cmac(pk_cmul(pk(a1), pk(b1)), pk(a2), pk(b2)));
    cplx c;
    c.re = -a1.im * b1.im + a1.re * b1.re - a2.im * b2.im + a2.re * b2.re;
    c.im = a1.re * b1.im + a1.im * b1.re + a2.re * b2.im + a2.im * b2.re;
    return c;
}

__device__ __forceinline__ su3 mul(const su3& a, const su3& b)
{
    su3 c;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            c.e[i][j] = dot3(a.e[i][0], a.e[i][1], a.e[i][2], b.e[0][j], b.e[1][j], b.e[2][j]);
    return c;
}
__device__ __forceinline__ su2 mul(const su2& a, const su2& b)
{
    su2 c;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) c.e[i][j] = dot2(a.e[i][0], a.e[i][1], b.e[0][j], b.e[1][j]);
    return c;
}

template <int N>
__device__ __forceinline__ mat<N> herm(const mat<N>& a)
{
    mat<N> r;
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
        for (int j = 0; j < N; j++) r.e[i][j] = conj(a.e[j][i]);
    return r;
}

template <int N>
__device__ __forceinline__ void add_to(mat<N>& a, const mat<N>& b)
{
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
        for (int j = 0; j < N; j++) a.e[i][j] = cadd(a.e[i][j], b.e[i][j]);
}

template <int N>
__device__ __forceinline__ mat<N> zero()
{
    mat<N> r;
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
        for (int j = 0; j < N; j++) r.e[i][j] = mk((real)0, (real)0);
    return r;
}

template <int N>
__device__ __forceinline__ mat<N> unity()
{
    mat<N> r = zero<N>();
#pragma unroll
    for (int i = 0; i < N; i++) r.e[i][i].re = (real)1;
    return r;
}

template <int N>
__device__ __forceinline__ cplx trace(const mat<N>& a)
{
    cplx t = a.e[0][0];
#pragma unroll
    for (int i = 1; i < N; i++) t = cadd(t, a.e[i][i]);
    return t;
}

// third row from the first two: w = (u x v)*  (SU(N)/suncpp/su3/algebra_su3.cpp:219-226)
__device__ __forceinline__ void reconstruct(su3& a)
{
    const cplx u1 = a.e[0][0], u2 = a.e[0][1], u3 = a.e[0][2];
    const cplx v1 = a.e[1][0], v2 = a.e[1][1], v3 = a.e[1][2];
    if constexpr (!REAL_IS_DOUBLE) {
        // w = conj(u x v): w1 = conj(u2 v3 - u3 v2), ...
        const v2f p1 = pk(u1), p2 = pk(u2), p3 = pk(u3), q1 = pk(v1), q2 = pk(v2), q3 = pk(v3);
        const v2f w1 = pk_cmac(pk_cmul(p2, q3), -p3, q2), w2 = pk_cmac(pk_cmul(p3, q1), -p1, q3), w3 = pk_cmac(pk_cmul(p1, q2), -p2, q1);
        a.e[2][0] = mk((real)w1.x, (real)-w1.y);
        a.e[2][1] = mk((real)w2.x, (real)-w2.y);
        a.e[2][2] = mk((real)w3.x, (real)-w3.y);
        return;
    }
    a.e[2][0].re = u3.im * v2.im - u3.re * v2.re - u2.im * v3.im + u2.re * v3.re;
    a.e[2][0].im = u3.re * v2.im + u3.im * v2.re - u2.re * v3.im - u2.im * v3.re;
    a.e[2][1].re = -u3.im * v1.im + u3.re * v1.re + u1.im * v3.im - u1.re * v3.re;
    a.e[2][1].im = -u3.re * v1.im - u3.im * v1.re + u1.re * v3.im + u1.im * v3.re;
    a.e[2][2].re = u2.im * v1.im - u2.re * v1.re - u1.im * v2.im + u1.re * v2.re;
    a.e[2][2].im = u2.re * v1.im + u2.im * v1.re - u1.re * v2.im - u1.im * v2.re;
}
// second row of an SU(2) element from the first (algebra_su2.cpp:137-142)
__device__ __forceinline__ void reconstruct(su2& a)
{
    a.e[1][0] = mk(-a.e[0][1].re, a.e[0][1].im);
    a.e[1][1] = mk(a.e[0][0].re, -a.e[0][0].im);
}

// Gram-Schmidt re-unitarisation (SU(N)/suncpp/su3/algebra_su3.cpp:185-217).  `m` supplies the
// square root and reciprocal (device libm or the hardware-seeded versions of fastmath.h).
template <class Math>
__device__ __forceinline__ void gram_schmidt(su3& a, const Math& m)
{
    cplx u1 = a.e[0][0], u2 = a.e[0][1], u3 = a.e[0][2];
    cplx v1 = a.e[1][0], v2 = a.e[1][1], v3 = a.e[1][2];
    // one reciprocal norm and six products instead of the reference's six quotients (<= 1 ulp apart)
    real norm_u, iu;
    m.sqrt_rcp(u1.re * u1.re + u2.re * u2.re + u3.re * u3.re + u1.im * u1.im + u2.im * u2.im + u3.im * u3.im, norm_u, iu);
    u1 = mk(u1.re * iu, u1.im * iu);
    u2 = mk(u2.re * iu, u2.im * iu);
    u3 = mk(u3.re * iu, u3.im * iu);
    const cplx sp = cadd(cadd(cmul(v1, conj(u1)), cmul(v2, conj(u2))), cmul(v3, conj(u3)));
    v1 = csub(v1, cmul(u1, sp));
    v2 = csub(v2, cmul(u2, sp));
    v3 = csub(v3, cmul(u3, sp));
    real norm_v, iv;
    m.sqrt_rcp(v1.re * v1.re + v2.re * v2.re + v3.re * v3.re + v1.im * v1.im + v2.im * v2.im + v3.im * v3.im, norm_v, iv);
    v1 = mk(v1.re * iv, v1.im * iv);
    v2 = mk(v2.re * iv, v2.im * iv);
    v3 = mk(v3.re * iv, v3.im * iv);
    a.e[0][0] = u1; a.e[0][1] = u2; a.e[0][2] = u3;
    a.e[1][0] = v1; a.e[1][1] = v2; a.e[1][2] = v3;
    reconstruct(a);
}
__device__ __forceinline__ void gram_schmidt(su3& a) { gram_schmidt(a, MathLib()); }
// SU(2): normalise the first row, rebuild the second (kernel lattice_GramSchmidt,
// SU(N)/suncl/suncl.cl:349-396 with lattice_su2_Normalize)
__device__ __forceinline__ void gram_schmidt(su2& a)
{
    cplx u1 = a.e[0][0], u2 = a.e[0][1];
    const real n = sqrt(u1.re * u1.re + u1.im * u1.im + u2.re * u2.re + u2.im * u2.im);
    a.e[0][0] = mk(u1.re / n, u1.im / n);
    a.e[0][1] = mk(u2.re / n, u2.im / n);
    reconstruct(a);
}

}  // namespace QCD_NS
