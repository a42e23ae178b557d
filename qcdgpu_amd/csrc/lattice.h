// lattice.h -- data layout of the link table in HBM and the index maps of the kernels.
//
// Site numbering is the reference's: gid = y + z*Ny + t*Ny*Nz + x*Ny*Nz*Nt, y fastest and x
// slowest (SU(N)/suncpp/coord_work/coord_work.cpp:40-61).  Sites 2h and 2h+1 differ in y only
// (Ny is even), so they have opposite parity and "half index" h = gid >> 1 numbers the sites
// of EITHER parity; it is also the reference's thread index of an update pass
// (coord_work.cpp:151-175).
//
// Layout (differs from the reference's 4-vector rows on purpose): even/odd split,
// component-major, one fp64 complex (16 B) per lane:
//
//     U[dir][parity][comp][h]      h in [0, vh_alloc),  comp in [0, NCOMP)      (one complex number of the library's precision)
//
// NCOMP = 6 for SU(3) (rows u and v; w = (u x v)* is rebuilt in registers), 2 for SU(2)
// (row u).  A wavefront reading component c of 64 consecutive half-sites moves 1 KiB with one
// global_load_dwordx4 per lane; a (parity, dir) pass streams its own parity's arrays linearly.
// A slab context (halo = 1) stores nx + 2 x-slices: slice 0 and slice nx + 1 are the halos.
#pragma once

#include <hip/hip_runtime.h>

#include "su_math.h"

// the C++ names of the library live in namespace QCD_NS: qcd, or qcd_f32 in the single-precision build, so that both
// libraries can be loaded into one process and a profile tells their kernels apart
#ifndef QCD_NS
#define QCD_NS qcd
#endif
namespace QCD_NS {

struct Geom {
    int nx, ny, nz, nt;  // owned (interior) extents; nx = number of owned x slices
    int halo;            // 0 or 1
    int x0;              // global x of owned slice 0
    int nyh;             // ny / 2
    int slice_h;         // half-sites per x slice = nyh*nz*nt
    int vh_int;          // owned half-sites = nx*slice_h
    int vh_alloc;        // stored half-sites = (nx + 2*halo)*slice_h
    int gsites;          // sites of the FULL lattice
    int gslice;          // sites per x slice = ny*nz*nt
    // update-pass work distribution over the 8 XCDs (each has its own 4 MiB L2): the (z,t)
    // plane is cut into nzr x ntr = 8 regions (or 1 = no cut); workgroup b works for region
    // b % 8 and walks that region's half-sites with x slowest, so that the workgroups an XCD
    // runs at one time cover a few consecutive x-slices of ONE (z,t) box and find their
    // +-y, +-z, +-t and +-x neighbours in that XCD's L2
    int nzr, ntr;        // regions along z and t
    int rz, rt;          // z and t extent of a region
    int region_h;        // half-sites of a region per x slice = nyh*rz*rt
    // when region_h, nyh*rz and nyh are all powers of two: their logarithms in bits 16.., 8..15, 0..7, so that a lane
    // finds its site with shifts; -1: by divisions (three per site and wavefront: 5 % of a 32^4 pass's instructions)
    int idx_shifts;
    // shape of the block of sites one workgroup updates inside its region: all of y times
    // tz x tt x tx sites in z, t, x (tz = 0: a run of consecutive half-sites, y then z then t)
    int tz, tt, tx;
    // x-slices an update launch works on: unx slices starting at owned slice ux0, wrapping around
    // the owned range (ux0 = 0, unx = nx: the whole slab).  Lets a slab update its boundary
    // slices first, so that their halo messages travel while the interior is updated.
    int ux0, unx;
    // bit d set: the extent in direction d (0 x, 2 z, 3 t; y is always even) is odd.  Across the periodic wrap of
    // such a direction a site's neighbour has the SAME parity; only the kernels that look a neighbour's parity
    // up per site (load_site_link) serve such a lattice, and a checkerboard pass runs in phases (update_phase)
    int odd;             // bit 4: twisted boundary condition (general kernels too, no phases of its own)
    // Random streams: `share` sites of one parity share a stream -- the sites (x, y-pair, z, t' + k tq), k = 0 .. share-1,
    // tq = nt / share, in this order.  A lane that updates them one after the other moves the generator state once
    // instead of `share` times.  Partners lie tq apart, not side by side, so that the blocks in flight at one time
    // still form a compact box of the (z, t) plane.  The regions above then cut the (z, t mod tq) plane: rt = tq / ntr.
    int share, tq;
    // twist (bit 4 of odd): U_y read at global x = Nx - 1 has its first row times (tw[0] + i tw[1]) and, SU(3), its second
    // row times (tw[2] + i tw[3]); the last row follows from the rows before it as for every stored link
    double tw[4];
};

struct Site {
    int xs, y, z, t;  // xs: owned-slice index; -1 and nx address the halos of a slab
};

template <int GROUP>
struct ncomp {
    static constexpr int value = (GROUP == 3) ? 6 : 2;
};

__device__ __forceinline__ int half_index(const Geom& g, const Site& s)
{
    return (((s.xs + g.halo) * g.nt + s.t) * g.nz + s.z) * g.nyh + (s.y >> 1);
}
__device__ __forceinline__ int site_parity(const Geom& g, const Site& s)
{
    return (g.x0 + s.xs + s.y + s.z + s.t) & 1;
}
// site of parity `par` with interior half index t (t in [0, vh_int))
__device__ __forceinline__ Site site_of(const Geom& g, int t, int par)
{
    Site s;
    s.xs = t / g.slice_h;
    int r = t - s.xs * g.slice_h;
    const int zt = g.nyh * g.nz;
    s.t = r / zt;
    r -= s.t * zt;
    s.z = r / g.nyh;
    const int yh = r - s.z * g.nyh;
    const int yb = (par + g.x0 + s.xs + s.z + s.t) & 1;
    s.y = 2 * yh + yb;
    return s;
}
// number of the random stream that serves the owned half-site (xs, t, z, yh), inside the owned slab, and the
// site's place in the stream's order of use (Geom::share)
__host__ __device__ __forceinline__ int stream_of(const Geom& g, int xs, int t, int z, int yh, int* sub = nullptr)
{
    const int k = t / g.tq;
    if (sub) *sub = k;
    return ((xs * g.tq + (t - k * g.tq)) * g.nz + z) * g.nyh + yh;
}
// site of parity `par` handled by thread `tid` of block `b` in an update pass; false if the thread has no
// site.  Block b belongs to region b % regions; T = b / regions counts the region's blocks: group T / share
// (a run of BLOCK half-sites of the region's (z, t mod tq) box, x slowest), member T % share of the group
// (which of the `share` partner sites).  *stream is the random stream of the site (stream_of); with share = 1
// it is the site pair's half index inside the owned slab, the reference's thread index
// (SU(N)/suncpp/coord_work/coord_work.cpp:151-175).
template <int BLOCK>
__device__ __forceinline__ bool site_of_workgroup(const Geom& g, int b, int tid, int par, Site& s, int& stream)
{
    const int regions = g.nzr * g.ntr;
    const int k = b % regions;
    int zl, tl, yh, sub = 0;
    if (g.tz > 0) {
        // compact block (share = 1 only): tiles run z fastest, then t, then x inside the region
        int T = b / regions;
        const int ntz = g.rz / g.tz, ntt = g.rt / g.tt;
        const int Tz = T % ntz;
        T /= ntz;
        const int Tt = T % ntt;
        const int Tx = T / ntt;
        if (Tx * g.tx >= g.unx) return false;
        yh = tid % g.nyh;
        int r = tid / g.nyh;
        zl = Tz * g.tz + r % g.tz;
        r /= g.tz;
        tl = Tt * g.tt + r % g.tt;
        s.xs = Tx * g.tx + r / g.tt;
    } else {
        const int T = b / regions;
        const int grp = T / g.share;
        sub = T - grp * g.share;
        const int w = grp * BLOCK + tid;
        if (w >= g.unx * g.region_h) return false;
        if (g.idx_shifts >= 0) {   // wave-uniform
            const int sr = g.idx_shifts >> 16, sp = (g.idx_shifts >> 8) & 255, sy = g.idx_shifts & 255;
            s.xs = w >> sr;
            int q = w & ((1 << sr) - 1);
            tl = q >> sp;
            q &= (1 << sp) - 1;
            zl = q >> sy;
            yh = q & ((1 << sy) - 1);
        } else {
            s.xs = w / g.region_h;
            int q = w - s.xs * g.region_h;
            const int plane = g.nyh * g.rz;
            tl = q / plane;
            q -= tl * plane;
            zl = q / g.nyh;
            yh = q - zl * g.nyh;
        }
    }
    s.xs += g.ux0;   // position inside the launch's slice range -> owned slice, wrapping
    if (s.xs >= g.nx) s.xs -= g.nx;
    s.z = (k % g.nzr) * g.rz + zl;
    const int tp = (k / g.nzr) * g.rt + tl;   // t mod tq
    s.t = tp + sub * g.tq;
    const int yb = (par + g.x0 + s.xs + s.z + s.t) & 1;
    s.y = 2 * yh + yb;
    stream = ((s.xs * g.tq + tp) * g.nz + s.z) * g.nyh + yh;
    return true;
}

// nearest neighbours with periodic wrap (coord_work.cpp:63-149); in a slab the x direction
// does not wrap: the halo slices stand in for the neighbouring contexts
__device__ __forceinline__ Site fwd(const Geom& g, Site s, int dir)
{
    switch (dir) {
        case 0: s.xs++; if (!g.halo && s.xs >= g.nx) s.xs = 0; break;
        case 1: s.y++; if (s.y >= g.ny) s.y = 0; break;
        case 2: s.z++; if (s.z >= g.nz) s.z = 0; break;
        default: s.t++; if (s.t >= g.nt) s.t = 0; break;
    }
    return s;
}
__device__ __forceinline__ Site bwd(const Geom& g, Site s, int dir)
{
    switch (dir) {
        case 0: if (!g.halo && s.xs <= 0) s.xs = g.nx; s.xs--; break;
        case 1: if (s.y <= 0) s.y = g.ny; s.y--; break;
        case 2: if (s.z <= 0) s.z = g.nz; s.z--; break;
        default: if (s.t <= 0) s.t = g.nt; s.t--; break;
    }
    return s;
}

template <int GROUP>
__host__ __device__ __forceinline__ size_t comp_base(const Geom& g, int dir, int par)
{
    return (size_t)((dir * 2 + par) * ncomp<GROUP>::value) * (size_t)g.vh_alloc;
}

__device__ __forceinline__ su3 load_link(const real2* __restrict__ U, const Geom& g, int dir,
                                         int par, int h, su3*)
{
    const real2* p = U + comp_base<3>(g, dir, par) + h;
    su3 m;
#pragma unroll
    for (int c = 0; c < 6; c++) {
        const real2 v = p[(size_t)c * g.vh_alloc];
        m.e[c / 3][c % 3] = mk(v.x, v.y);
    }
    reconstruct(m);
    return m;
}
__device__ __forceinline__ su2 load_link(const real2* __restrict__ U, const Geom& g, int dir,
                                         int par, int h, su2*)
{
    const real2* p = U + comp_base<2>(g, dir, par) + h;
    su2 m;
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const real2 v = p[(size_t)c * g.vh_alloc];
        m.e[0][c] = mk(v.x, v.y);
    }
    reconstruct(m);
    return m;
}
// Omega U for the links the twisted boundary condition applies to (SU(N)/suncl/su3_matrix_memory.cl:57-98,
// su2_matrix_memory.cl:63-88): rows times phases, then the dependent row again
__device__ __forceinline__ void twist_link(su3& m, const Geom& g)
{
    const cplx p1 = mk((real)g.tw[0], (real)g.tw[1]), p2 = mk((real)g.tw[2], (real)g.tw[3]);
#pragma unroll
    for (int j = 0; j < 3; j++) {
        m.e[0][j] = cmul(m.e[0][j], p1);
        m.e[1][j] = cmul(m.e[1][j], p2);
    }
    reconstruct(m);
}
__device__ __forceinline__ void twist_link(su2& m, const Geom& g)
{
    const cplx p1 = mk((real)g.tw[0], (real)g.tw[1]);
    m.e[0][0] = cmul(m.e[0][0], p1);
    m.e[0][1] = cmul(m.e[0][1], p1);
    reconstruct(m);
}
// a link as updates and measurements see it
template <int GROUP>
__device__ __forceinline__ mat<GROUP> load_site_link(const real2* __restrict__ U, const Geom& g,
                                                     int dir, const Site& s)
{
    mat<GROUP> m = load_link(U, g, dir, site_parity(g, s), half_index(g, s), (mat<GROUP>*)nullptr);
    if ((g.odd & 16) && dir == 1 && g.x0 + s.xs == g.gsites / g.gslice - 1) twist_link(m, g);
    return m;
}

__device__ __forceinline__ void store_link(real2* __restrict__ U, const Geom& g, int dir, int par,
                                           int h, const su3& m)
{
    real2* p = U + comp_base<3>(g, dir, par) + h;
#pragma unroll
    for (int c = 0; c < 6; c++) p[(size_t)c * g.vh_alloc] = make_real2(m.e[c / 3][c % 3].re, m.e[c / 3][c % 3].im);
}
__device__ __forceinline__ void store_link(real2* __restrict__ U, const Geom& g, int dir, int par,
                                           int h, const su2& m)
{
    real2* p = U + comp_base<2>(g, dir, par) + h;
#pragma unroll
    for (int c = 0; c < 2; c++) p[(size_t)c * g.vh_alloc] = make_real2(m.e[0][c].re, m.e[0][c].im);
}

using u32x4 = unsigned int __attribute__((ext_vector_type(4)));
using u32x2 = unsigned int __attribute__((ext_vector_type(2)));

// one stored complex number per lane: 16 bytes (fp64) or 8 bytes (fp32); AUX = cache policy bits
template <int AUX>
__device__ __forceinline__ void buf_load_c(double2& v, __amdgpu_buffer_rsrc_t r, int voffset, int soffset)
{
    v = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(r, voffset, soffset, AUX));
}
template <int AUX>
__device__ __forceinline__ void buf_load_c(float2& v, __amdgpu_buffer_rsrc_t r, int voffset, int soffset)
{
    v = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(r, voffset, soffset, AUX));
}
template <int AUX = 0>
__device__ __forceinline__ real2 buf_load_c(__amdgpu_buffer_rsrc_t r, int voffset, int soffset)
{
    real2 v;
    buf_load_c<AUX>(v, r, voffset, soffset);
    return v;
}
template <int AUX = 0>
__device__ __forceinline__ void buf_store_c(double2 v, __amdgpu_buffer_rsrc_t r, int voffset, int soffset)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voffset, soffset, AUX);
}
template <int AUX = 0>
__device__ __forceinline__ void buf_store_c(float2 v, __amdgpu_buffer_rsrc_t r, int voffset, int soffset)
{
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, voffset, soffset, AUX);
}

// ---------------------------------------------------------------------------------------------
// Buffer-descriptor access for the update pass.  One descriptor per (direction, parity)
// sub-array (wave-uniform, built by scalar instructions); the lane's half index goes into the
// 32-bit vector offset and the component stride into the scalar offset, so a link costs one
// integer add per lane instead of six 64-bit address computations.

template <int GROUP>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t link_rsrc(const real2* U, const Geom& g, int dir, int par)
{
    return __builtin_amdgcn_make_buffer_rsrc((void*)(U + comp_base<GROUP>(g, dir, par)), 0,
                                             ncomp<GROUP>::value * g.vh_alloc * CB, 0x00020000);
}

// both parities of one direction behind one descriptor: the second parity's links start
// ncomp * vh_alloc * 16 bytes behind the first's (for kernels whose waves hold sites of both)
template <int GROUP>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t link_rsrc_both(const real2* U, const Geom& g, int dir)
{
    return __builtin_amdgcn_make_buffer_rsrc((void*)(U + comp_base<GROUP>(g, dir, 0)), 0,
                                             2 * ncomp<GROUP>::value * g.vh_alloc * CB, 0x00020000);
}

// byte_off = CB * half index
__device__ __forceinline__ su3 load_link_buf(__amdgpu_buffer_rsrc_t r, const Geom& g, int byte_off, su3*)
{
    su3 m;
#pragma unroll
    for (int c = 0; c < 6; c++) {
        const real2 v = buf_load_c(r, byte_off, c * g.vh_alloc * CB);
        m.e[c / 3][c % 3] = mk(v.x, v.y);
    }
    reconstruct(m);
    return m;
}
__device__ __forceinline__ su2 load_link_buf(__amdgpu_buffer_rsrc_t r, const Geom& g, int byte_off, su2*)
{
    su2 m;
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const real2 v = buf_load_c(r, byte_off, c * g.vh_alloc * CB);
        m.e[0][c] = mk(v.x, v.y);
    }
    reconstruct(m);
    return m;
}
// cache policy of link stores: 2 = non-temporal.  A pass does not read the links it writes, and the
// 4 MiB L2 of an XCD is far smaller than what its waves have in flight, so stored lines should not
// push out lines that a neighbour site is about to ask for (-1.5 % on the 32^4 update pass)
#ifndef QCD_STORE_AUX
#define QCD_STORE_AUX 2
#endif
template <int AUX = QCD_STORE_AUX>
__device__ __forceinline__ void store_link_buf(__amdgpu_buffer_rsrc_t r, const Geom& g, int byte_off, const su3& m)
{
#pragma unroll
    for (int c = 0; c < 6; c++)
        buf_store_c<AUX>(make_real2(m.e[c / 3][c % 3].re, m.e[c / 3][c % 3].im), r, byte_off, c * g.vh_alloc * CB);
}
// cache policy sc0 | sc1: the store is written through to memory, so that once the wavefront's
// store count is back at zero a kernel that starts on any XCD reads the new value
constexpr int STORE_AUX_THROUGH = 1 | 16;
__device__ __forceinline__ void store_link_buf(__amdgpu_buffer_rsrc_t r, const Geom& g, int byte_off, const su2& m)
{
#pragma unroll
    for (int c = 0; c < 2; c++)
        buf_store_c<QCD_STORE_AUX>(make_real2(m.e[0][c].re, m.e[0][c].im), r, byte_off, c * g.vh_alloc * CB);
}

// byte offsets from a site's half index to its eight nearest neighbours (periodic wrap
// included; no wrap in x for a slab, whose halo slices are stored in line)
struct NbrOff {
    int f0, f1, f2, f3;  // to x + dir
    int b0, b1, b2, b3;  // to x - dir
};
__device__ __forceinline__ NbrOff neighbour_offsets(const Geom& g, const Site& s)
{
    NbrOff n;
    const int sx = g.slice_h * CB, sz = g.nyh * CB, st = g.nyh * g.nz * CB;
    n.f0 = (!g.halo && s.xs == g.nx - 1) ? -(g.nx - 1) * sx : sx;
    n.b0 = (!g.halo && s.xs == 0) ? (g.nx - 1) * sx : -sx;
    const int yb = s.y & 1;
    n.f1 = (s.y == g.ny - 1) ? -(g.nyh - 1) * CB : (yb ? CB : 0);
    n.b1 = (s.y == 0) ? (g.nyh - 1) * CB : (yb ? 0 : -CB);
    n.f2 = (s.z == g.nz - 1) ? -(g.nz - 1) * sz : sz;
    n.b2 = (s.z == 0) ? (g.nz - 1) * sz : -sz;
    n.f3 = (s.t == g.nt - 1) ? -(g.nt - 1) * st : st;
    n.b3 = (s.t == 0) ? (g.nt - 1) * st : -st;
    return n;
}
// dir is wave-uniform: the selects are on a scalar condition
__device__ __forceinline__ int off_fwd(const NbrOff& n, int dir)
{
    return dir == 0 ? n.f0 : (dir == 1 ? n.f1 : (dir == 2 ? n.f2 : n.f3));
}
__device__ __forceinline__ int off_bwd(const NbrOff& n, int dir)
{
    return dir == 0 ? n.b0 : (dir == 1 ? n.b1 : (dir == 2 ? n.b2 : n.b3));
}

}  // namespace QCD_NS
