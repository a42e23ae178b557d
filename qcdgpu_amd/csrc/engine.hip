// engine.hip -- C-ABI of libqcdgpu_amd.so (declared in include/qcdgpu_amd.h) over the kernels.
//
// A context owns one x-slab of the lattice on one device: the link table, the per-thread RANLUX
// streams, the closed-form "gid" table when INTS = 2, and the reduction buffers.  Every entry
// point validates its arguments against the geometry the kernels were sized for before anything
// is launched, and reports failures as status codes.
#include "../../include/qcdgpu_amd.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <mutex>
#include <vector>

#include "host_seed.h"
#include "kernels.h"

using namespace QCD_NS;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(QCD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                               \
    } while (0)

// sites per workgroup of the update pass: one wavefront (64) and four (256) run the 32^4 pass in
// 165.6 us, two (128) in 168.4, eight (512) in 204
#ifndef QCD_UPD_BLOCK
#define QCD_UPD_BLOCK 64
#endif
constexpr int UPD_BLOCK = QCD_UPD_BLOCK;
#ifndef QCD_MEAS_BLOCK
#define QCD_MEAS_BLOCK 256
#endif
constexpr int MEAS_BLOCK = QCD_MEAS_BLOCK;   // threads per workgroup of the measurement kernels
constexpr size_t DRAWS_BYTES = (size_t)DRAW_SLOTS * DRAW_SLOT_STRIDE * sizeof(unsigned long long);

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Workspace of qcd_measure_plaquette_sums_static, the one entry point that allocates nothing: the
// per-workgroup partial sums of the plaquette kernel live in this array of the code object instead
// of in memory the context allocated.  Every device the library runs on holds its own copy until
// the process ends (1 MiB).  The name is a host-side stand-in, not a device address: host code uses
// the per-device address that hipGetSymbolAddress returned (g_scratch_ptr), never the name.
constexpr int STATIC_PARTIALS_MAX = 65536;   // workgroups of MEAS_BLOCK sites: lattices up to 2^24 sites per context
static __device__ __align__(256) double2 scratch[STATIC_PARTIALS_MAX];
constexpr int STATIC_MAX_DEVICES = 64;
double2* g_scratch_ptr[STATIC_MAX_DEVICES];   // device address of `scratch` on each device, 0 = not asked yet

// address of this device's copy of `scratch`; the device must be current.  Called from qcd_create,
// so that the entry point itself finds the address ready.
int static_scratch_setup(int device)
{
    // (a device beyond the table is served by every entry point but the one that needs the workspace, which says so)
    if (device < 0 || device >= STATIC_MAX_DEVICES) return QCD_OK;
    static std::mutex mu;   // contexts may be created from several host threads
    std::lock_guard<std::mutex> lk(mu);
    if (g_scratch_ptr[device]) return QCD_OK;
    void* p = nullptr;
    HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    g_scratch_ptr[device] = (double2*)p;
    return QCD_OK;
}

}  // namespace

// error text of the other translation units of the library (qcd_last_error returns it)
namespace QCD_NS {
int fail_msg(int code, const char* msg)
{
    g_err = msg;
    return code;
}
}  // namespace QCD_NS

struct qcd_ctx {
    qcd_params p;
    real2* edge_buf;          // message buffer of qcd_halo_copy (lazily allocated)
    hipEvent_t edge_ev;
    Geom g;
    int ncomp;
    int upd_grid;             // workgroups per update launch (0: one per block of sites)
    int gh_grid;              // workgroups (one per CU) of the gather / heat-bath pair kernel (0: not available)
    // qcd_update_signal_edges: what the next update launch signals (slices = 0: nothing), and the
    // device word its edge blocks count in
    struct { int slices; unsigned* flag; unsigned seq; } edge_sig = {0, nullptr, 0u};
    unsigned* edge_count = nullptr;
    int gh_mode;              // 0 never, 1 when the launch has at least 8 * gh_grid blocks, 2 whenever possible
    unsigned gh_passes;       // whole-lattice passes run as the pair kernel so far (its parity picks the walking direction)
    hipStream_t own_stream, stream;
    real2* U;
    size_t U_elems;
    RlxView rv;
    unsigned char* slice_slot;   // per owned x-slice: the slot its streams' states are in (see RlxView::cur)
    float* gidr;
    double2* partial;
    int npartial;
    long partial_len;         // double2 elements allocated for partial
    double2* loops;           // Polyakov loop (Re, Im) of every owned space site, index y + Ny (z + Nz xs)
    bool loops_fresh;         // loops belongs to the present links: the last call on the context wrote it
    double2* site_vals;       // plaquette sums (space-like, time-like) of every owned site; allocated by the
    bool sites_fresh;         // first plaquette profile, then kept up by every plaquette measurement
    int result_len;           // doubles in result_host
    double* result_host;      // pinned, coherent
    double* result_host_dev;  // the same memory as the device addresses it
    unsigned long long* draws_dev;   // DRAW_SLOTS partial counts, DRAW_SLOT_STRIDE elements apart
    unsigned long long* draws_host;  // pinned, same shape
    // side lane of qcd_update_split: the rest of a pass runs on its own stream, next to the
    // boundary part on the context's stream
    hipStream_t side_stream;
    hipEvent_t fork_ev, join_ev;
    bool side_pending;        // the context's stream has not yet waited for the side lane's last launch
    bool initialized;
    bool profiling;
    hipEvent_t ev0, ev1;
    qcd_timing timing;
};

namespace {

inline int regions_of(const Geom& g) { return g.nzr * g.ntr; }

int set_device(const qcd_ctx* c)
{
    HIP_TRY(hipSetDevice(c->p.device));
    return QCD_OK;
}

struct Timed {
    qcd_ctx* c;
    int kind;
    Timed(qcd_ctx* ctx, int k) : c(ctx), kind(k)
    {
        if (c->profiling) (void)hipEventRecord(c->ev0, c->stream);
    }
    int finish()
    {
        unsigned long long* n = kind == 0 ? &c->timing.update_launches
                              : kind == 1 ? &c->timing.measure_launches : &c->timing.halo_launches;
        double* ms = kind == 0 ? &c->timing.update_ms : kind == 1 ? &c->timing.measure_ms : &c->timing.halo_ms;
        (*n)++;
        HIP_TRY(hipGetLastError());
        if (c->profiling) {
            HIP_TRY(hipEventRecord(c->ev1, c->stream));
            HIP_TRY(hipEventSynchronize(c->ev1));
            float t = 0.f;
            HIP_TRY(hipEventElapsedTime(&t, c->ev0, c->ev1));
            *ms += t;
        }
        return QCD_OK;
    }
};

// whether a heat-bath pass over xs_count owned slices runs as the gather / heat-bath pair kernel (k_update_gh): SU(3),
// per-thread streams, even extents, whole blocks of sites and enough of them for the kernel's teams
bool pass_is_paired(const qcd_ctx* c, int xs_count)
{
    const Geom& g = c->g;
    if (c->p.group != 3 || c->p.start_mode == QCD_START_GID || UPD_BLOCK != 64 || g.odd || g.tz != 0 || c->gh_grid <= 0) return false;
    if (((long)xs_count * g.region_h) % UPD_BLOCK != 0) return false;   // the kernel has no code for partly filled blocks
    // shared streams need the kernel's teams to be pairs of wavefronts: a heat-bath wavefront then takes every block of its team
    if (g.share > 1 && GH_TEAMS != GH_SIDE) return false;
    const int ngroups = g.nzr * g.ntr * (int)cdiv((long)xs_count * g.region_h, UPD_BLOCK);
    const int nblocks = ngroups * g.share;
    return c->gh_mode == 2 || (c->gh_mode == 1 && nblocks >= 2 * GH_SIDE * c->gh_grid && ngroups >= GH_SIDE * c->gh_grid);
}

// NSKIP: 223 = RANLUX3 (the reference's default, QCDGPU.cpp:98) with the skip unrolled at compile
// time; 0 = any luxury level, skip count at run time
template <int GROUP, bool GID, int NSKIP>
int launch_update_dir(qcd_ctx* c, int parity, int dir, int xs_begin, int xs_count, bool side = false)
{
    hipStream_t stream = side ? c->side_stream : c->stream;
    Geom g = c->g;
    g.ux0 = xs_begin;
    g.unx = xs_count;
    const int regions = g.nzr * g.ntr;
    const int ngroups = regions * (int)cdiv((long)g.unx * g.region_h, UPD_BLOCK);   // groups of `share` blocks (Geom::share)
    const int nblocks = ngroups * g.share;
    // launch no more workgroups than the device holds at one time and let each walk several
    // groups of blocks: a finished wave goes straight on to its next site instead of waiting for
    // its stores to drain and for the dispatcher to start a successor
    int nwg = ngroups;
    if (c->upd_grid > 0 && c->upd_grid < ngroups) nwg = c->upd_grid;
    const dim3 grid(nwg), block(UPD_BLOCK);
    const double beta_eff = c->p.beta / c->p.group;  // SU(N)/suncpp/Update/sun_update.h:94
    UpdArgs a;
    a.U = c->U;
    a.g = g;
    a.parity = parity;
    a.nhit = c->p.nhit;
    a.beta_eff = beta_eff;
    a.beta_inv = 1.0 / beta_eff;
    a.rv = c->rv;
    if (!GID) {
        // the slices of the range must all have been through the same number of passes
        const int s0 = c->slice_slot[xs_begin % g.nx];
        for (int k = 1; k < xs_count; k++)
            if (c->slice_slot[(xs_begin + k) % g.nx] != s0)
                return fail(QCD_ERR_STATE, "update of slices %d..%d: some of them have been updated one pass more often than others", xs_begin,
                            xs_begin + xs_count - 1);
        a.rv.cur = s0;
    }
    a.gidr = c->gidr;
    a.draws = c->draws_dev;
    a.nblocks = nblocks;
    a.ngroups = ngroups;
    a.phase = -1;
    a.reverse = 0;
    a.edge_local = 0;
    a.edge_total = 0u;
    a.edge_seq = 0u;
    a.edge_count = nullptr;
    a.edge_flag = nullptr;
    const int edge_slices = c->edge_sig.slices;
    c->edge_sig.slices = 0;
    bool paired = false;   // launched as the gather / heat-bath pair kernel
    if (g.odd) {
        // phases by the number of odd directions in which a site sits at the far end (update_phase): at most three
        if (edge_slices > 0 || xs_count != g.nx) return fail(QCD_ERR_UNSUPPORTED, "lattices with an odd extent are updated in whole passes");
        const int nph = 1 + ((g.odd & 1) && dir != 0) + ((g.odd & 4) && dir != 2) + ((g.odd & 8) && dir != 3);
        for (a.phase = 0; a.phase < nph; a.phase++) {
#define QCD_LAUNCH_ANY(D) hipLaunchKernelGGL((k_update<GROUP, D, GID, (GID ? 0 : NSKIP), UPD_BLOCK, true>), grid, block, 0, stream, a)
            switch (dir) {
                case 0: QCD_LAUNCH_ANY(0); break;
                case 1: QCD_LAUNCH_ANY(1); break;
                case 2: QCD_LAUNCH_ANY(2); break;
                default: QCD_LAUNCH_ANY(3); break;
            }
#undef QCD_LAUNCH_ANY
        }
        HIP_TRY(hipGetLastError());
        if (!GID)
            for (int k = 0; k < xs_count; k++) c->slice_slot[(xs_begin + k) % g.nx] ^= 1;
        return QCD_OK;
    }
    if constexpr (GROUP == 3 && !GID && UPD_BLOCK == 64) {
        if (pass_is_paired(c, xs_count)) {
            paired = true;
            if (edge_slices > 0) {
                // the launch starts at the last `edge_slices` slices and wraps to the first ones: with
                // whole blocks per slice the first blocks of every region are exactly those slices
                if (g.region_h % UPD_BLOCK != 0 || !c->edge_count)
                    return fail(QCD_ERR_UNSUPPORTED, "edge signal: a slice of this lattice is not a whole number of blocks per region");
                a.edge_local = 2 * edge_slices * (int)(g.region_h / UPD_BLOCK) * g.share;
                a.edge_total = (unsigned)a.edge_local * (unsigned)regions;
                a.edge_seq = c->edge_sig.seq;
                a.edge_count = c->edge_count;
                a.edge_flag = c->edge_sig.flag;
            }
            // every other whole-lattice pass walks the x-slices downwards (no edge signal: its edge blocks must come first)
            static const int rev_env = getenv("QCDGPU_AMD_REVERSE") ? atoi(getenv("QCDGPU_AMD_REVERSE")) : 1;   // diagnostic override: 0 never
            if (rev_env && edge_slices == 0 && xs_count == g.nx) a.reverse = (int)((c->gh_passes++ + (rev_env == 2 ? 1u : 0u)) & 1u);   // 2: the other passes
            const dim3 ggrid(c->gh_grid), gblock(512);
#define QCD_LAUNCH_GH(D) hipLaunchKernelGGL((k_update_gh<D, NSKIP, GH_TEAMS>), ggrid, gblock, 0, stream, a)
            switch (dir) {
                case 0: QCD_LAUNCH_GH(0); break;
                case 1: QCD_LAUNCH_GH(1); break;
                case 2: QCD_LAUNCH_GH(2); break;
                default: QCD_LAUNCH_GH(3); break;
            }
#undef QCD_LAUNCH_GH
        }
    }
    if (edge_slices > 0 && !paired)
        return fail(QCD_ERR_UNSUPPORTED, "edge signal: this pass does not run as the kernel that has it (SU(3), per-thread streams, at least %d blocks)",
                    2 * GH_SIDE * c->gh_grid);
#define QCD_LAUNCH_UPDATE(D) \
    hipLaunchKernelGGL((k_update<GROUP, D, GID, NSKIP, UPD_BLOCK>), grid, block, 0, stream, a)
    if (!paired) switch (dir) {
        case 0: QCD_LAUNCH_UPDATE(0); break;
        case 1: QCD_LAUNCH_UPDATE(1); break;
        case 2: QCD_LAUNCH_UPDATE(2); break;
        default: QCD_LAUNCH_UPDATE(3); break;
    }
#undef QCD_LAUNCH_UPDATE
    HIP_TRY(hipGetLastError());
    // the launch is in the stream: from now on the slices' generator states are in the other slot
    if (!GID)
        for (int k = 0; k < xs_count; k++) c->slice_slot[(xs_begin + k) % g.nx] ^= 1;
    return QCD_OK;
}

template <int GROUP>
int launch_overrelax(qcd_ctx* c, int parity, int dir)
{
    const Geom& g = c->g;
    const dim3 grid(g.nzr * g.ntr * cdiv((long)g.nx * g.region_h, UPD_BLOCK) * g.share), block(UPD_BLOCK);
    if (g.odd) {
        const int nph = 1 + ((g.odd & 1) && dir != 0) + ((g.odd & 4) && dir != 2) + ((g.odd & 8) && dir != 3);
        for (int ph = 0; ph < nph; ph++) switch (dir) {
            case 0: hipLaunchKernelGGL((k_overrelax<GROUP, 0, UPD_BLOCK, true>), grid, block, 0, c->stream, c->U, g, parity, ph); break;
            case 1: hipLaunchKernelGGL((k_overrelax<GROUP, 1, UPD_BLOCK, true>), grid, block, 0, c->stream, c->U, g, parity, ph); break;
            case 2: hipLaunchKernelGGL((k_overrelax<GROUP, 2, UPD_BLOCK, true>), grid, block, 0, c->stream, c->U, g, parity, ph); break;
            default: hipLaunchKernelGGL((k_overrelax<GROUP, 3, UPD_BLOCK, true>), grid, block, 0, c->stream, c->U, g, parity, ph); break;
        }
        return QCD_OK;
    }
    if constexpr (GROUP == 3 && UPD_BLOCK == 64) {
        // a pass with enough blocks runs through the gather / second-wavefront pairs of the heat-bath pass (k_update_gh)
        static const bool ovr_gh = !(getenv("QCDGPU_AMD_OVR_GH") && atoi(getenv("QCDGPU_AMD_OVR_GH")) == 0);   // diagnostic override: 0 never
        if (ovr_gh && pass_is_paired(c, g.nx)) {
            UpdArgs a = {};
            a.U = c->U;
            a.g = g;
            a.g.ux0 = 0;
            a.g.unx = g.nx;
            a.parity = parity;
            a.ngroups = g.nzr * g.ntr * (int)cdiv((long)g.nx * g.region_h, UPD_BLOCK);
            a.nblocks = a.ngroups * g.share;
            a.phase = -1;
            const dim3 ggrid(c->gh_grid), gblock(512);
            switch (dir) {
                case 0: hipLaunchKernelGGL((k_update_gh<0, 0, GH_TEAMS, true>), ggrid, gblock, 0, c->stream, a); break;
                case 1: hipLaunchKernelGGL((k_update_gh<1, 0, GH_TEAMS, true>), ggrid, gblock, 0, c->stream, a); break;
                case 2: hipLaunchKernelGGL((k_update_gh<2, 0, GH_TEAMS, true>), ggrid, gblock, 0, c->stream, a); break;
                default: hipLaunchKernelGGL((k_update_gh<3, 0, GH_TEAMS, true>), ggrid, gblock, 0, c->stream, a); break;
            }
            return QCD_OK;
        }
    }
    switch (dir) {
        case 0: hipLaunchKernelGGL((k_overrelax<GROUP, 0, UPD_BLOCK>), grid, block, 0, c->stream, c->U, g, parity, -1); break;
        case 1: hipLaunchKernelGGL((k_overrelax<GROUP, 1, UPD_BLOCK>), grid, block, 0, c->stream, c->U, g, parity, -1); break;
        case 2: hipLaunchKernelGGL((k_overrelax<GROUP, 2, UPD_BLOCK>), grid, block, 0, c->stream, c->U, g, parity, -1); break;
        default: hipLaunchKernelGGL((k_overrelax<GROUP, 3, UPD_BLOCK>), grid, block, 0, c->stream, c->U, g, parity, -1); break;
    }
    return QCD_OK;
}

int launch_update(qcd_ctx* c, int parity, int dir, int xs_begin, int xs_count, bool side = false)
{
    const bool gid = (c->p.start_mode == QCD_START_GID);
    const bool lux3 = (c->p.lux_level == 3);
    Timed tm(c, 0);
    int rc;
    if (c->p.group == 3) {
        if (gid) rc = launch_update_dir<3, true, 0>(c, parity, dir, xs_begin, xs_count, side);
        else if (lux3) rc = launch_update_dir<3, false, 223>(c, parity, dir, xs_begin, xs_count, side);
        else rc = launch_update_dir<3, false, 0>(c, parity, dir, xs_begin, xs_count, side);
    } else {
        if (gid) rc = launch_update_dir<2, true, 0>(c, parity, dir, xs_begin, xs_count, side);
        else if (lux3) rc = launch_update_dir<2, false, 223>(c, parity, dir, xs_begin, xs_count, side);
        else rc = launch_update_dir<2, false, 0>(c, parity, dir, xs_begin, xs_count, side);
    }
    if (rc) return rc;
    return tm.finish();
}

// closed-form pseudo-random numbers of INTS = 2 (SU(N)/suncpp/su2/update_su2.cpp:71-76, 95-98),
// evaluated with the host's libm in the reference's precision mix: double argument, the value
// cast to float.  gid is the GLOBAL site number, fsites the FULL volume.
void gid_randoms(int gid, int nhit, int fsites, float out[6])
{
    out[0] = fabsf((float)sin((0.005 * (1.0 + nhit) + 270.0 / fsites) * gid));
    out[1] = fabsf((float)cos((0.005 * (1.0 + nhit) + 60.0 / fsites) * gid));
    out[2] = fabsf((float)sin((0.005 * (1.0 + nhit) - 150.0 / fsites) * gid));
    out[3] = fabsf((float)cos((0.005 * (1.0 + nhit) - 380.0 / fsites) * gid));
    out[4] = fabsf((float)cos((0.08 - 270.0 / fsites) * gid));
    out[5] = fabsf((float)sin((0.08 - 60.0 / fsites) * gid));
}

// gid start, one link in the reference device layout (lattice_matrixGID,
// SU(N)/suncpp/su3/algebra_su3.cpp:64-80 incl. its Gram-Schmidt; su2/algebra_su2.cpp:54-64)
void gid_link(int group, int gid, int dir, int sites, double q[12])
{
    const double a = (dir + 1) / 1000.0;
    if (group == 3) {
        double u[6], v[6];
        u[0] = sin((a + 10.0 / sites) * gid) / 2.0;  u[1] = cos((a - 10.0 / sites) * gid) / 3.0;
        u[2] = sin((a - 18.0 / sites) * gid) / 4.0;  u[3] = cos((a + 12.0 / sites) * gid) / 4.0;
        u[4] = sin((a - 12.0 / sites) * gid) / 3.0;  u[5] = cos((a - 15.0 / sites) * gid) / 2.0;
        v[0] = sin((a + 15.0 / sites) * gid) / 3.0;  v[1] = cos((a - 11.0 / sites) * gid) / 4.0;
        v[2] = sin((a - 11.0 / sites) * gid) / 3.0;  v[3] = cos((a + 16.0 / sites) * gid) / 2.0;
        v[4] = sin((a - 16.0 / sites) * gid) / 3.0;  v[5] = cos((a - 18.0 / sites) * gid) / 2.0;
        // Gram-Schmidt on the host, same operation order as the device/CPU twin
        const double nu = sqrt(u[0] * u[0] + u[2] * u[2] + u[4] * u[4] + u[1] * u[1] + u[3] * u[3] + u[5] * u[5]);
        for (int k = 0; k < 6; k++) u[k] = u[k] / nu;
        // sp = sum_k v_k * conj(u_k)
        double spr = 0.0, spi = 0.0;
        {
            double tr[3], ti[3];
            for (int k = 0; k < 3; k++) {
                const double vr = v[2 * k], vi = v[2 * k + 1], ur = u[2 * k], ui = -u[2 * k + 1];
                tr[k] = vr * ur - vi * ui;
                ti[k] = vr * ui + vi * ur;
            }
            spr = (tr[0] + tr[1]) + tr[2];
            spi = (ti[0] + ti[1]) + ti[2];
        }
        for (int k = 0; k < 3; k++) {
            const double ur = u[2 * k], ui = u[2 * k + 1];
            const double pr = ur * spr - ui * spi, pi = ur * spi + ui * spr;
            v[2 * k] = v[2 * k] - pr;
            v[2 * k + 1] = v[2 * k + 1] - pi;
        }
        const double nv = sqrt(v[0] * v[0] + v[2] * v[2] + v[4] * v[4] + v[1] * v[1] + v[3] * v[3] + v[5] * v[5]);
        for (int k = 0; k < 6; k++) v[k] = v[k] / nv;
        // quads: (Re u1,Re u2,Re u3,Re v3), (Im u1,Im u2,Im u3,Im v3), (Re v1,Re v2,Im v1,Im v2)
        q[0] = u[0]; q[1] = u[2]; q[2] = u[4]; q[3] = v[4];
        q[4] = u[1]; q[5] = u[3]; q[6] = u[5]; q[7] = v[5];
        q[8] = v[0]; q[9] = v[2]; q[10] = v[1]; q[11] = v[3];
    } else {
        const double u1re = sin((a + 10.0 / sites) * gid) / 2.0;
        const double u2im = cos((a - 10.0 / sites) * gid) / 3.0;
        const double u2re = sin((a - 18.0 / sites) * gid) / 4.0;
        const double u1im = sqrt(1 - u1re * u1re - u2im * u2im - u2re * u2re);
        q[0] = u1re; q[1] = u2re; q[2] = u1im; q[3] = u2im;
    }
}

int convert_slices(qcd_ctx* c, real4* dev_table, int rowsize, int slice0, int nslices, bool to_engine)
{
    const int n = nslices * c->g.gslice;
    const dim3 grid(cdiv(n, MEAS_BLOCK)), block(MEAS_BLOCK);
    if (c->p.group == 3) {
        if (to_engine) hipLaunchKernelGGL((k_convert<3, true, MEAS_BLOCK>), grid, block, 0, c->stream, c->U, c->g, dev_table, rowsize, slice0, nslices);
        else hipLaunchKernelGGL((k_convert<3, false, MEAS_BLOCK>), grid, block, 0, c->stream, c->U, c->g, dev_table, rowsize, slice0, nslices);
    } else {
        if (to_engine) hipLaunchKernelGGL((k_convert<2, true, MEAS_BLOCK>), grid, block, 0, c->stream, c->U, c->g, dev_table, rowsize, slice0, nslices);
        else hipLaunchKernelGGL((k_convert<2, false, MEAS_BLOCK>), grid, block, 0, c->stream, c->U, c->g, dev_table, rowsize, slice0, nslices);
    }
    HIP_TRY(hipGetLastError());
    return QCD_OK;
}

// host table of `nslices` slices (global x of the first: gx0, periodic) in the reference layout
int upload_gid_slices(qcd_ctx* c, int gx0, int slice0, int nslices)
{
    const Geom& g = c->g;
    const int rows = (c->p.group == 3) ? 12 : 4;
    const int quads = rows / 4;
    const int n = nslices * g.gslice;
    std::vector<real> tab((size_t)rows * n * 4);
    for (int xl = 0; xl < nslices; xl++) {
        const int gx = ((gx0 + xl) % c->p.L[0] + c->p.L[0]) % c->p.L[0];
        for (int r = 0; r < g.gslice; r++) {
            const int gid = gx * g.gslice + r;
            const int lg = xl * g.gslice + r;
            for (int dir = 0; dir < 4; dir++) {
                double q[12];
                gid_link(c->p.group, gid, dir, g.gsites, q);
                for (int k = 0; k < quads; k++)
                    for (int e = 0; e < 4; e++) tab[((size_t)(k * 4 + dir) * n + lg) * 4 + e] = (real)q[4 * k + e];
            }
        }
    }
    real4* dev = nullptr;
    HIP_TRY(hipMalloc(&dev, tab.size() * sizeof(real)));
    hipError_t e = hipMemcpyAsync(dev, tab.data(), tab.size() * sizeof(real), hipMemcpyHostToDevice, c->stream);
    int rc = QCD_OK;
    if (e != hipSuccess) rc = fail(QCD_ERR_HIP, "hipMemcpyAsync: %s", hipGetErrorString(e));
    if (rc == QCD_OK) rc = convert_slices(c, dev, n, slice0, nslices, true);
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(dev);
    return rc;
}

int seed_streams(qcd_ctx* c)
{
    const Geom& g = c->g;
    const int ns = c->rv.ns;
    const int nskip = rlx_nskip_for_level(c->p.lux_level);
    // stream index = GLOBAL number of the stream (stream_of with the global x; with share = 1 the global half
    // index of the thread's site pair): results do not depend on how the lattice is cut into slabs
    const long first = (long)c->p.x0 * (g.slice_h / g.share);
    std::vector<int> seeds = stream_seeds(c->p.rand_series, first, ns);
    std::vector<unsigned> st((size_t)RLX_SLOT_WORDS * ns);
    for (int i = 0; i < ns; i++) {
        RlxRegs r;
        unsigned d[RLX_SLOT_WORDS];
        rlx_seed_state(seeds[(size_t)i], nskip, r);
        rlx_pack(r, d);
        for (int k = 0; k < RLX_SLOT_WORDS; k++) st[rlx_slot_index(k, (size_t)ns, (size_t)i)] = d[k];
    }
    HIP_TRY(hipMemcpyAsync(c->rv.slot0, st.data(), st.size() * sizeof(unsigned), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->rv.slot1, 0, st.size() * sizeof(unsigned), c->stream));
    memset(c->slice_slot, 0, (size_t)g.nx);
    HIP_TRY(hipMemsetAsync(c->draws_dev, 0, DRAWS_BYTES, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->rv.q = nskip / 24;
    c->rv.rem = nskip % 24;
    return QCD_OK;
}

// join = false: the call may run beside the side lane of qcd_update_split (halo messages only)
// reads_links: the call leaves the links as they are (measurements), so the per-site arrays of the
// profile measurements stay valid; every other call invalidates them
int check_ctx(qcd_ctx* c, bool need_init, bool join = true, bool reads_links = false)
{
    if (!c) return fail(QCD_ERR_INVALID, "null context");
    if (!reads_links) c->loops_fresh = c->sites_fresh = false;
    if (need_init && !c->initialized) return fail(QCD_ERR_STATE, "lattice not initialised: call qcd_init_lattice or qcd_upload first");
    int rc = set_device(c);
    if (rc) return rc;
    if (join && c->side_pending) {
        HIP_TRY(hipStreamWaitEvent(c->stream, c->join_ev, 0));
        c->side_pending = false;
    }
    return QCD_OK;
}

// total of the partial draw counts
int read_draws(qcd_ctx* c, unsigned long long* total)
{
    HIP_TRY(hipMemcpyAsync(c->draws_host, c->draws_dev, DRAWS_BYTES, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    unsigned long long t = 0;
    for (int k = 0; k < DRAW_SLOTS; k++) t += c->draws_host[(size_t)k * DRAW_SLOT_STRIDE];
    *total = t;
    return QCD_OK;
}

int reduce_partials(qcd_ctx* c, int n, int nout, double* host_out)
{
    // the reduce kernel writes the few result words straight into pinned host memory: no copy command
    // between the kernel and the synchronisation (about 10 us per measurement call)
    hipLaunchKernelGGL((k_reduce<MEAS_BLOCK>), dim3(nout), dim3(MEAS_BLOCK), 0, c->stream, c->partial, n, c->result_host_dev);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int i = 0; i < 2 * nout; i++) host_out[i] = c->result_host[i];
    return QCD_OK;
}

}  // namespace

extern "C" {

const char* qcd_last_error(void) { return g_err.c_str(); }

int qcd_precision(void) { return REAL_IS_DOUBLE ? 2 : 1; }

int qcd_create(qcd_ctx** out, const qcd_params* p)
{
    if (!out || !p) return fail(QCD_ERR_INVALID, "null argument");
    *out = nullptr;
    if (p->group != 2 && p->group != 3) return fail(QCD_ERR_INVALID, "GROUP must be 2 or 3 (got %d)", p->group);
    if (p->nd != 4) return fail(QCD_ERR_UNSUPPORTED, "ND must be 4 (got %d)", p->nd);
    // The precision is the library's, fixed when it was built (the reference fixes it when it builds its kernels,
    // -D PRECISION, SU(N)/suncl/suncl.cpp:3274-3301): 2 = double; 1 = single and 3 = mixed, which the reference's
    // kernels do not tell apart (float links and arithmetic, fp64 sums)
    if (REAL_IS_DOUBLE ? (p->precision != 2) : (p->precision != 1 && p->precision != 3))
        return fail(QCD_ERR_UNSUPPORTED, "this library is built for %s precision (qcd_precision() = %d), the context asks for %d",
                    REAL_IS_DOUBLE ? "double" : "single", qcd_precision(), p->precision);
    for (int i = 0; i < 4; i++)
        if (p->L[i] < 2) return fail(QCD_ERR_INVALID, "lattice extent L[%d] = %d < 2", i, p->L[i]);
    if (p->L[1] & 1) return fail(QCD_ERR_INVALID, "Ny = %d must be even (checkerboard pairs sites 2h, 2h+1)", p->L[1]);
    // Odd Nx, Nz or Nt (the reference needs only Ny even, SU(N)/suncpp/coord_work/coord_work.cpp:151-175): across the
    // periodic wrap of an odd extent a site's neighbour has the SAME parity.  One context that owns the whole lattice
    // serves this with kernels that look every neighbour's parity up and with passes run in phases (update_phase in
    // kernels.h); a slab with halos does not (the wrap neighbour lives in another context, inside the same pass)
    if (((p->L[0] | p->L[2] | p->L[3]) & 1) && p->halo)
        return fail(QCD_ERR_UNSUPPORTED, "odd lattice extents (%d x %d x %d x %d) are supported for one uncut lattice only, not for x-slabs", p->L[0],
                    p->L[1], p->L[2], p->L[3]);
    if ((p->phi != 0.0 || p->omega != 0.0) && (p->halo || ((p->L[0] | p->L[2] | p->L[3]) & 1)))
        return fail(QCD_ERR_UNSUPPORTED, "twisted boundary conditions (PHI, OMEGA) are supported for one uncut lattice with even extents only");
    if (p->rng_share != 0 && p->rng_share != 1 && p->rng_share != 2 && p->rng_share != 4)
        return fail(QCD_ERR_INVALID, "rng_share = %d is not 0 (automatic), 1, 2 or 4", p->rng_share);
    if (p->rng_share > 1 && p->L[3] % p->rng_share)
        return fail(QCD_ERR_INVALID, "rng_share = %d does not divide Nt = %d", p->rng_share, p->L[3]);
    if (p->nhit < 1) return fail(QCD_ERR_INVALID, "NHIT = %d < 1", p->nhit);
    if (p->lux_level < 0 || p->lux_level > 4) return fail(QCD_ERR_INVALID, "RANLUX level %d not in 0..4", p->lux_level);
    if (p->rand_series == 0) return fail(QCD_ERR_INVALID, "RANDSERIES = 0 (time seed) must be resolved by the host");
    if (p->start_mode != QCD_START_HOT && p->start_mode != QCD_START_COLD && p->start_mode != QCD_START_GID)
        return fail(QCD_ERR_INVALID, "start mode %d is not hot (0), cold (1) or gid (2)", p->start_mode);
    if (p->nx_local < 1 || p->x0 < 0 || p->x0 + p->nx_local > p->L[0])
        return fail(QCD_ERR_INVALID, "slab [%d, %d) outside 0..%d", p->x0, p->x0 + p->nx_local, p->L[0]);
    if (p->halo != 0 && p->halo != 1) return fail(QCD_ERR_INVALID, "halo must be 0 or 1");
    if (!p->halo && (p->x0 != 0 || p->nx_local != p->L[0]))
        return fail(QCD_ERR_INVALID, "halo = 0 requires the context to own the whole lattice in x");
    const long sites = (long)p->L[0] * p->L[1] * p->L[2] * p->L[3];
    if (sites > (1L << 30)) return fail(QCD_ERR_UNSUPPORTED, "lattice of %ld sites exceeds 2^30", sites);
    {
        // the kernels address a direction's links through 32-bit byte offsets (buffer descriptors)
        const long vh = (long)(p->nx_local + 2 * p->halo) * (p->L[1] / 2) * p->L[2] * p->L[3];
        const long bytes = 2L * (p->group == 3 ? 6 : 2) * vh * CB;
        if (bytes >= (1L << 31))
            return fail(QCD_ERR_UNSUPPORTED, "%ld bytes of links per direction on one device exceed 2^31: cut the lattice into more parts", bytes);
    }

    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (p->device < 0 || p->device >= ndev) return fail(QCD_ERR_INVALID, "device %d of %d", p->device, ndev);
    HIP_TRY(hipSetDevice(p->device));
    {
        const int rc_scratch = static_scratch_setup(p->device);
        if (rc_scratch) return rc_scratch;
    }

    qcd_ctx* c = new qcd_ctx();
    memset(c, 0, sizeof(*c));
    c->p = *p;
    Geom& g = c->g;
    g.nx = p->nx_local; g.ny = p->L[1]; g.nz = p->L[2]; g.nt = p->L[3];
    g.halo = p->halo;
    g.x0 = p->x0;
    g.nyh = g.ny / 2;
    g.slice_h = g.nyh * g.nz * g.nt;
    g.vh_int = g.nx * g.slice_h;
    g.vh_alloc = (g.nx + 2 * g.halo) * g.slice_h;
    g.gsites = (int)sites;
    g.gslice = g.ny * g.nz * g.nt;
    g.odd = (p->L[0] & 1) | ((p->L[2] & 1) << 2) | ((p->L[3] & 1) << 3);
    g.tw[0] = g.tw[2] = 1.0;
    g.tw[1] = g.tw[3] = 0.0;
    if (p->phi != 0.0 || p->omega != 0.0) {
        // SU(3): rows times e^{i (phi + omega) / 2} and e^{i (omega - phi) / 2} (su3_matrix_memory.cl:66-95); SU(2): e^{i phi / 2}
        g.odd |= 16;
        const double t1 = p->group == 3 ? 0.5 * (p->phi + p->omega) : 0.5 * p->phi, t2 = 0.5 * (p->omega - p->phi);
        g.tw[0] = cos(t1); g.tw[1] = sin(t1);
        g.tw[2] = cos(t2); g.tw[3] = sin(t2);
    }
    // Sites per random stream (qcd_params::rng_share).  Automatic: a function of the FULL lattice only, so that the
    // parts of a cut lattice agree.  Sharing divides the number of independent units of work of a pass, so it is for
    // lattices whose pass keeps the device busy even then: at least 2048 groups of 64 sites per pass.
    g.share = p->rng_share;
    if (g.share == 0) {
        g.share = (g.odd & 16) ? 1 : qcd_rng_share_for(p->L, p->L[0]);
        if (const char* ov = getenv("QCDGPU_AMD_SHARE")) {   // diagnostic override of the automatic choice
            const int v = atoi(ov);
            if ((v == 1 || v == 2 || v == 4) && p->L[3] % v == 0) g.share = v;
        }
        c->p.rng_share = g.share;
    }
    g.tq = g.nt / g.share;
    // (z,t) regions for the 8 XCDs: the split with the smallest cut surface that divides both
    // extents; none -> one region (plain linear order).  With shared streams the plane that is cut is (z, t mod tq),
    // and its t extent is a cut in any case (t = tq - 1 and t = tq belong to different members of a group)
    g.nzr = g.ntr = 1;
    if (!getenv("QCDGPU_AMD_LINEAR_ORDER")) {  // diagnostic switch: plain linear workgroup order
        static const int cand[4][2] = {{2, 4}, {4, 2}, {1, 8}, {8, 1}};
        double best = 1e30;
        for (auto& cd : cand) {
            if (g.nz % cd[0] || g.tq % cd[1]) continue;
            const int rz = g.nz / cd[0], rt = g.tq / cd[1];
            // fraction of a region's sites that sit on a cut face
            const double cut = (cd[0] > 1 ? 1.0 / rz : 0.0) + ((cd[1] > 1 || g.share > 1) ? 1.0 / rt : 0.0);
            if (cut < best) { best = cut; g.nzr = cd[0]; g.ntr = cd[1]; }
        }
    }
    if (const char* ov = getenv("QCDGPU_AMD_REGIONS")) {  // diagnostic override "nzr,ntr"
        int a = 0, b = 0;
        if (sscanf(ov, "%d,%d", &a, &b) == 2 && a > 0 && b > 0 && g.nz % a == 0 && g.tq % b == 0) {
            g.nzr = a;
            g.ntr = b;
        }
    }
    g.rz = g.nz / g.nzr;
    g.rt = g.tq / g.ntr;
    g.region_h = g.nyh * g.rz * g.rt;   // half-sites of a region per x slice and group member
    {
        auto log2_of = [](int v) { int l = 0; while ((1 << l) < v) l++; return ((1 << l) == v) ? l : -1; };
        const int lr = log2_of(g.region_h), lp = log2_of(g.nyh * g.rz), ly = log2_of(g.nyh);
        g.idx_shifts = (lr >= 0 && lp >= 0 && ly >= 0 && !getenv("QCDGPU_AMD_NO_SHIFTS")) ? ((lr << 16) | (lp << 8) | ly) : -1;   // diagnostic switch: always divide
    }
    g.tz = g.tt = g.tx = 0;
    g.ux0 = 0;
    g.unx = g.nx;
    if (const char* ov = getenv("QCDGPU_AMD_TILE")) {  // diagnostic override "tz,tt,tx"
        int a = 0, b = 0, d = 0;
        if (sscanf(ov, "%d,%d,%d", &a, &b, &d) == 3 && a > 0 && b > 0 && d > 0 && UPD_BLOCK % g.nyh == 0 &&
            a * b * d == UPD_BLOCK / g.nyh && g.rz % a == 0 && g.rt % b == 0 && g.nx % d == 0 && g.share == 1) {
            g.tz = a;
            g.tt = b;
            g.tx = d;
        }
    }
    // update launches: as many workgroups as the device holds at one time (a multiple of the
    // region count, so that a workgroup stays inside one region)
    c->upd_grid = 0;
    {
        hipDeviceProp_t prop;
        int per_cu = 0;
        const void* fn = (p->group == 3) ? (const void*)k_update<3, 0, false, 223, UPD_BLOCK>
                                         : (const void*)k_update<2, 0, false, 223, UPD_BLOCK>;
        if (hipGetDeviceProperties(&prop, p->device) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, UPD_BLOCK, 0) == hipSuccess && per_cu > 0) {
            const int resident = per_cu * prop.multiProcessorCount;
            c->upd_grid = resident / regions_of(g) * regions_of(g);
        }
    }
    c->gh_grid = 0;
    c->gh_mode = 1;
    if (p->group == 3) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, p->device) == hipSuccess)
            c->gh_grid = prop.multiProcessorCount / regions_of(g) * regions_of(g);
    }
    if (const char* ov = getenv("QCDGPU_AMD_GH")) c->gh_mode = atoi(ov);              // diagnostic override: 0, 1 or 2
    if (const char* ov = getenv("QCDGPU_AMD_GRID")) {  // diagnostic override: workgroups per update launch
        const int v = atoi(ov);                        // (0: one workgroup per block of sites)
        c->upd_grid = v > 0 ? (v + regions_of(g) - 1) / regions_of(g) * regions_of(g) : 0;
    }
    c->ncomp = (p->group == 3) ? 6 : 2;
    c->U_elems = (size_t)8 * c->ncomp * g.vh_alloc;
    c->npartial = cdiv(2L * g.vh_int, MEAS_BLOCK);

    int rc = QCD_OK;
    auto tryhip = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && rc == QCD_OK) rc = fail(e == hipErrorOutOfMemory ? QCD_ERR_NOMEM : QCD_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    };
    tryhip(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking), "hipStreamCreate");
    c->stream = c->own_stream;
    tryhip(hipEventCreate(&c->ev0), "hipEventCreate");
    tryhip(hipEventCreate(&c->ev1), "hipEventCreate");
    tryhip(hipMalloc(&c->U, c->U_elems * sizeof(real2)), "hipMalloc(links)");
    const size_t nstreams = (size_t)g.vh_int / g.share;
    const size_t stw = (size_t)RLX_SLOT_WORDS * nstreams * sizeof(unsigned);
    tryhip(hipMalloc(&c->rv.slot0, 2 * stw), "hipMalloc(ranlux)");   // both slots, back to back
    c->rv.slot1 = c->rv.slot0 ? c->rv.slot0 + (size_t)RLX_SLOT_WORDS * nstreams : nullptr;
    c->slice_slot = (unsigned char*)calloc((size_t)g.nx, 1);
    if (!c->slice_slot && rc == QCD_OK) rc = fail(QCD_ERR_NOMEM, "calloc(slice slots)");
    c->rv.ns = (int)nstreams;
    if (p->start_mode == QCD_START_GID)
        tryhip(hipMalloc(&c->gidr, (size_t)6 * g.nx * g.gslice * sizeof(float)), "hipMalloc(gid table)");
    // the profile measurements need one pair per plane and up to 64 partials per plane
    const int max_ext = g.nx > g.ny ? (g.nx > g.nz ? g.nx : g.nz) : (g.ny > g.nz ? g.ny : g.nz);
    c->partial_len = 6L * c->npartial > 64L * max_ext ? 6L * c->npartial : 64L * max_ext;
    c->result_len = 2 * max_ext > 16 ? 2 * max_ext : 16;
    tryhip(hipMalloc(&c->partial, (size_t)c->partial_len * sizeof(double2)), "hipMalloc(partials)");
    tryhip(hipMalloc(&c->loops, (size_t)g.nx * g.ny * g.nz * sizeof(double2)), "hipMalloc(polyakov loops)");
    tryhip(hipHostMalloc(&c->result_host, c->result_len * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent), "hipHostMalloc(result)");
    if (rc == QCD_OK) tryhip(hipHostGetDevicePointer((void**)&c->result_host_dev, c->result_host, 0), "hipHostGetDevicePointer(result)");
    tryhip(hipMalloc(&c->draws_dev, DRAWS_BYTES), "hipMalloc(draws)");
    if (g.halo) {   // slabs only: the counter of qcd_update_signal_edges, the side lane of qcd_update_split
        tryhip(hipMalloc((void**)&c->edge_count, sizeof(unsigned)), "hipMalloc(edge counter)");
        if (c->edge_count) tryhip(hipMemset(c->edge_count, 0, sizeof(unsigned)), "hipMemset(edge counter)");
        // (the side stream itself is made by the first qcd_update_split: a process has few hardware queues,
        // four by default, and streams beyond that share them -- what shares a queue runs in turn)
        tryhip(hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming), "hipEventCreate");
        tryhip(hipEventCreateWithFlags(&c->join_ev, hipEventDisableTiming), "hipEventCreate");
    }
    tryhip(hipHostMalloc(&c->draws_host, DRAWS_BYTES), "hipHostMalloc(draws)");
    if (rc != QCD_OK) {
        qcd_destroy(c);
        return rc;
    }
    *out = c;
    return QCD_OK;
}

void qcd_destroy(qcd_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->p.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    (void)hipFree(c->U);
    (void)hipFree(c->rv.slot0);
    free(c->slice_slot);
    (void)hipFree(c->gidr);
    (void)hipFree(c->edge_buf);
    if (c->edge_ev) (void)hipEventDestroy(c->edge_ev);
    (void)hipFree(c->partial);
    (void)hipFree(c->loops);
    (void)hipFree(c->site_vals);
    (void)hipFree(c->draws_dev);
    if (c->edge_count) (void)hipFree(c->edge_count);
    if (c->side_stream) {
        (void)hipStreamSynchronize(c->side_stream);
        (void)hipStreamDestroy(c->side_stream);
    }
    if (c->fork_ev) (void)hipEventDestroy(c->fork_ev);
    if (c->join_ev) (void)hipEventDestroy(c->join_ev);
    if (c->result_host) (void)hipHostFree(c->result_host);
    if (c->draws_host) (void)hipHostFree(c->draws_host);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int qcd_set_stream(qcd_ctx* c, void* hip_stream)
{
    int rc = check_ctx(c, false);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->stream = (hipStream_t)hip_stream;  // NULL is the device's default stream
    return QCD_OK;
}

int qcd_reset_stream(qcd_ctx* c)
{
    int rc = check_ctx(c, false);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->stream = c->own_stream;
    return QCD_OK;
}

int qcd_get_stream(qcd_ctx* c, void** hip_stream)
{
    if (!c || !hip_stream) return fail(QCD_ERR_INVALID, "null argument");
    *hip_stream = (void*)c->stream;
    return QCD_OK;
}

int qcd_get_params(const qcd_ctx* c, qcd_params* out)
{
    if (!c || !out) return fail(QCD_ERR_INVALID, "null argument");
    *out = c->p;
    return QCD_OK;
}

int qcd_synchronize(qcd_ctx* c)
{
    int rc = check_ctx(c, false);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return QCD_OK;
}

int qcd_init_lattice(qcd_ctx* c)
{
    int rc = check_ctx(c, false);
    if (rc) return rc;
    const Geom& g = c->g;
    if (c->p.start_mode == QCD_START_COLD || c->p.start_mode == QCD_START_HOT) {
        // hot start: unit links first, so that the halo slices of a slab hold group elements
        // until the first exchange brings the neighbours' links
        const dim3 grid(cdiv((long)c->U_elems, MEAS_BLOCK) > 4096 ? 4096 : cdiv((long)c->U_elems, MEAS_BLOCK)), block(MEAS_BLOCK);
        if (c->p.group == 3) hipLaunchKernelGGL((k_init_cold<3, MEAS_BLOCK>), grid, block, 0, c->stream, c->U, g);
        else hipLaunchKernelGGL((k_init_cold<2, MEAS_BLOCK>), grid, block, 0, c->stream, c->U, g);
        HIP_TRY(hipGetLastError());
    } else {
        // links of the owned slices and, for a slab, of the two halo slices
        rc = upload_gid_slices(c, c->p.x0 - g.halo, 0, g.nx + 2 * g.halo);
        if (rc) return rc;
        // closed-form random numbers of the owned sites
        const size_t n = (size_t)g.nx * g.gslice;
        std::vector<float> tab(6 * n);
        for (size_t lg = 0; lg < n; lg++) {
            float r[6];
            gid_randoms((int)(lg + (size_t)c->p.x0 * g.gslice), c->p.nhit, g.gsites, r);
            for (int k = 0; k < 6; k++) tab[k * n + lg] = r[k];
        }
        HIP_TRY(hipMemcpyAsync(c->gidr, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    rc = seed_streams(c);
    if (rc) return rc;
    if (c->p.start_mode == QCD_START_HOT) {
        const dim3 grid(cdiv((long)c->rv.ns, MEAS_BLOCK)), block(MEAS_BLOCK);
        if (c->p.group == 3) hipLaunchKernelGGL((k_init_hot<3, MEAS_BLOCK>), grid, block, 0, c->stream, c->U, g, c->rv, c->draws_dev);
        else hipLaunchKernelGGL((k_init_hot<2, MEAS_BLOCK>), grid, block, 0, c->stream, c->U, g, c->rv, c->draws_dev);
        HIP_TRY(hipGetLastError());
    }
    // the caller may switch streams right after this call: leave nothing in flight
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->initialized = true;
    return QCD_OK;
}

int qcd_rng_share_for(const int L[4], int min_nx_local)
{
    if (!L || min_nx_local < 1 || ((L[0] | L[2] | L[3]) & 1)) return 1;
    // sharing divides the number of independent units of work of a pass, so it is for parts whose pass keeps the
    // device busy even then: at least 2048 groups of 64 sites per pass and part
    const long blocks = (long)min_nx_local * (L[1] / 2) * L[2] * L[3] / 64;
    if (L[3] % 4 == 0 && blocks / 4 >= 2048) return 4;
    if (blocks / 2 >= 2048) return 2;
    return 1;
}

int qcd_update_kernel(qcd_ctx* c, int xs_count, int* pair_kernel)
{
    int rc = check_ctx(c, false, false, true);
    if (rc) return rc;
    if (!pair_kernel || xs_count < 1 || xs_count > c->g.nx) return fail(QCD_ERR_INVALID, "qcd_update_kernel: %d slices of %d", xs_count, c->g.nx);
    *pair_kernel = pass_is_paired(c, xs_count) ? 1 : 0;
    return QCD_OK;
}

int qcd_update(qcd_ctx* c, int parity, int dir)
{
    int rc = check_ctx(c, true);
    if (rc) return rc;
    if (parity != 0 && parity != 1) return fail(QCD_ERR_INVALID, "parity %d", parity);
    if (dir < 0 || dir > 3) return fail(QCD_ERR_INVALID, "direction %d", dir);
    return launch_update(c, parity, dir, 0, c->g.nx);
}

int qcd_update_range(qcd_ctx* c, int parity, int dir, int xs_begin, int xs_count)
{
    int rc = check_ctx(c, true);
    if (rc) return rc;
    if (parity != 0 && parity != 1) return fail(QCD_ERR_INVALID, "parity %d", parity);
    if (dir < 0 || dir > 3) return fail(QCD_ERR_INVALID, "direction %d", dir);
    if (xs_begin < 0 || xs_begin >= c->g.nx || xs_count < 1 || xs_count > c->g.nx)
        return fail(QCD_ERR_INVALID, "slice range %d + %d of %d owned slices", xs_begin, xs_count, c->g.nx);
    return launch_update(c, parity, dir, xs_begin, xs_count);
}

int qcd_update_signal_edges(qcd_ctx* c, int parity, int dir, int edge, void* signal, unsigned value)
{
    int rc = check_ctx(c, true);
    if (rc) return rc;
    if (parity != 0 && parity != 1) return fail(QCD_ERR_INVALID, "parity %d", parity);
    if (dir < 0 || dir > 3) return fail(QCD_ERR_INVALID, "direction %d", dir);
    const int nx = c->g.nx;
    if (!c->g.halo) return fail(QCD_ERR_STATE, "context has no halo slices");
    if (edge < 1 || 2 * edge >= nx) return fail(QCD_ERR_INVALID, "%d edge slices per face of %d owned slices", edge, nx);
    if (!signal) return fail(QCD_ERR_INVALID, "null signal");
    if (c->profiling) return fail(QCD_ERR_UNSUPPORTED, "edge signal: not with per-launch profiling");
    c->edge_sig.slices = edge;
    c->edge_sig.flag = (unsigned*)signal;
    c->edge_sig.seq = value;
    rc = launch_update(c, parity, dir, nx - edge, nx);   // the whole slab, starting at the last `edge` slices
    c->edge_sig.slices = 0;
    return rc;
}

int qcd_update_split(qcd_ctx* c, int parity, int dir, int first_begin, int first_count)
{
    int rc = check_ctx(c, true);
    if (rc) return rc;
    if (parity != 0 && parity != 1) return fail(QCD_ERR_INVALID, "parity %d", parity);
    if (dir < 0 || dir > 3) return fail(QCD_ERR_INVALID, "direction %d", dir);
    const int nx = c->g.nx;
    if (first_begin < 0 || first_begin >= nx || first_count < 1 || first_count > nx)
        return fail(QCD_ERR_INVALID, "slice range %d + %d of %d owned slices", first_begin, first_count, nx);
    if (first_count == nx) return launch_update(c, parity, dir, first_begin, first_count);
    const int rest_begin = (first_begin + first_count) % nx, rest_count = nx - first_count;
    if (c->g.halo && !c->side_stream && !c->profiling) HIP_TRY(hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
    if (!c->side_stream || c->profiling) {
        // no side lane (a context without halos), or per-launch timing asked for: one after the other
        rc = launch_update(c, parity, dir, first_begin, first_count);
        return rc ? rc : launch_update(c, parity, dir, rest_begin, rest_count);
    }
    // the rest of the pass needs everything issued so far, but not the first part
    HIP_TRY(hipEventRecord(c->fork_ev, c->stream));
    rc = launch_update(c, parity, dir, first_begin, first_count);
    if (rc) return rc;
    HIP_TRY(hipStreamWaitEvent(c->side_stream, c->fork_ev, 0));
    rc = launch_update(c, parity, dir, rest_begin, rest_count, true);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(c->join_ev, c->side_stream));
    c->side_pending = true;
    return QCD_OK;
}

int qcd_overrelax(qcd_ctx* c, int parity, int dir)
{
    int rc = check_ctx(c, true);
    if (rc) return rc;
    if (parity != 0 && parity != 1) return fail(QCD_ERR_INVALID, "parity %d", parity);
    if (dir < 0 || dir > 3) return fail(QCD_ERR_INVALID, "direction %d", dir);
    // the reference stores an updated link without taking the twist off again: a reflection stored that way is not microcanonical
    if (c->g.odd & 16) return fail(QCD_ERR_UNSUPPORTED, "overrelaxation with twisted boundary conditions");
    Timed tm(c, 0);
    rc = (c->p.group == 3) ? launch_overrelax<3>(c, parity, dir) : launch_overrelax<2>(c, parity, dir);
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return tm.finish();
}

int qcd_overrelax_sweep(qcd_ctx* c, int n)
{
    if (n < 0) return fail(QCD_ERR_INVALID, "n = %d", n);
    for (int s = 0; s < n; s++)
        for (int parity = 1; parity >= 0; parity--)
            for (int dir = 0; dir < 4; dir++) {
                const int rc = qcd_overrelax(c, parity, dir);
                if (rc) return rc;
            }
    return QCD_OK;
}

int qcd_sweep(qcd_ctx* c, int n)
{
    int rc = check_ctx(c, true);
    if (rc) return rc;
    if (c->g.halo) return fail(QCD_ERR_STATE, "qcd_sweep on a slab context: the halo exchange between passes is the caller's");
    for (int s = 0; s < n; s++) {
        for (int dir = 0; dir < 4; dir++)
            if ((rc = launch_update(c, 1, dir, 0, c->g.nx))) return rc;
        for (int dir = 0; dir < 4; dir++)
            if ((rc = launch_update(c, 0, dir, 0, c->g.nx))) return rc;
    }
    return QCD_OK;
}

int qcd_reunitarize(qcd_ctx* c)
{
    int rc = check_ctx(c, true);
    if (rc) return rc;
    Timed tm(c, 1);
    const dim3 grid(cdiv(2L * c->g.vh_int, MEAS_BLOCK)), block(MEAS_BLOCK);
    if (c->p.group == 3) hipLaunchKernelGGL((k_reunit<3, MEAS_BLOCK>), grid, block, 0, c->stream, c->U, c->g);
    else hipLaunchKernelGGL((k_reunit<2, MEAS_BLOCK>), grid, block, 0, c->stream, c->U, c->g);
    return tm.finish();
}

// the plaquette kernel with its per-workgroup partial sums going to `partial`; *nparts = how many
// Slices per thread for the kernels that march along x: as long as the marching threads still fill the device
static int march_segment(const Geom& g)
{
    const long fill = 40000;
    if ((long)g.gslice * cdiv(g.nx, 8) >= fill) return 8;
    if ((long)g.gslice * cdiv(g.nx, 4) >= fill) return 4;
    return 1;
}

static int launch_plaquette_into(qcd_ctx* c, double2* partial, double2* site_vals, int* nparts)
{
    const dim3 block(MEAS_BLOCK);
    if (c->p.group == 3) {
        // x segments per thread: long enough that the links carried along x pay, short enough for the grid to fill the device
        static const int seg_env = getenv("QCDGPU_AMD_PLAQ_XSEG") ? atoi(getenv("QCDGPU_AMD_PLAQ_XSEG")) : 0;   // diagnostic override (1: one site per thread)
        const Geom& g = c->g;
        // (a call at 16^4: 14 us with one site per thread, 28 us with four; 24^4: 65 / 69 / 62 us with 1 / 4 / 8; 32^4: 156 / 145 / 141 us)
        const int xseg = g.odd ? 1 : (seg_env ? seg_env : march_segment(g));
        if (xseg != 8 && xseg != 4) {
            hipLaunchKernelGGL((k_plaquette<3, MEAS_BLOCK>), dim3(c->npartial), block, 0, c->stream, c->U, c->g, partial, site_vals);
            *nparts = c->npartial;
            HIP_TRY(hipGetLastError());
            return QCD_OK;
        }
        const int nb = cdiv((long)g.gslice * cdiv(g.nx, xseg), MEAS_BLOCK);
        if (nb > c->npartial) return fail(QCD_ERR_STATE, "partial buffer too small");
        if (xseg == 8) hipLaunchKernelGGL((k_plaquette_march<MEAS_BLOCK, 8>), dim3(nb), block, 0, c->stream, c->U, c->g, partial, site_vals);
        else hipLaunchKernelGGL((k_plaquette_march<MEAS_BLOCK, 4>), dim3(nb), block, 0, c->stream, c->U, c->g, partial, site_vals);
        *nparts = nb;
    } else {
        hipLaunchKernelGGL((k_plaquette<2, MEAS_BLOCK>), dim3(c->npartial), block, 0, c->stream, c->U, c->g, partial, site_vals);
        *nparts = c->npartial;
    }
    HIP_TRY(hipGetLastError());
    return QCD_OK;
}

static int launch_plaquette(qcd_ctx* c, int* nparts) { return launch_plaquette_into(c, c->partial, c->site_vals, nparts); }

int qcd_measure_plaquette_sums(qcd_ctx* c, double sums[2])
{
    int rc = check_ctx(c, true, true, true);
    if (rc) return rc;
    if (!sums) return fail(QCD_ERR_INVALID, "null output");
    Timed tm(c, 1);
    int nparts = 0;
    if ((rc = launch_plaquette(c, &nparts))) return rc;
    if ((rc = tm.finish())) return rc;
    if ((rc = reduce_partials(c, nparts, 1, sums))) return rc;
    c->sites_fresh = c->site_vals != nullptr;
    return QCD_OK;
}

// The plaquette sums without any allocation, synchronisation or host copy: the kernels of
// qcd_measure_plaquette_sums with their partial sums in the code object's own `scratch` array and
// the two result words stored to memory of the caller (device-accessible).  Asynchronous on the
// context's stream.  Every call on a device shares that device's array: the plaquette kernel
// writes all the partials the reduce kernel reads, so nothing has to be cleared between calls,
// but calls that overlap on one device (two streams, two host threads) race.
int qcd_measure_plaquette_sums_static(qcd_ctx* c, double* sums_dev)
{
    int rc = check_ctx(c, true, true, true);
    if (rc) return rc;
    if (!sums_dev) return fail(QCD_ERR_INVALID, "null output");
    if (c->npartial > STATIC_PARTIALS_MAX)
        return fail(QCD_ERR_UNSUPPORTED, "%d workgroups of partial sums exceed the static workspace of %d (at most %ld owned sites)",
                    c->npartial, STATIC_PARTIALS_MAX, (long)STATIC_PARTIALS_MAX * MEAS_BLOCK);
    if (c->p.device >= STATIC_MAX_DEVICES)
        return fail(QCD_ERR_UNSUPPORTED, "the static workspace is kept for devices 0..%d; this context runs on device %d", STATIC_MAX_DEVICES - 1, c->p.device);
    double2* ws = g_scratch_ptr[c->p.device];
    if (!ws) return fail(QCD_ERR_STATE, "no static workspace on device %d", c->p.device);
    int nparts = 0;
    if ((rc = launch_plaquette_into(c, ws, nullptr, &nparts))) return rc;
    hipLaunchKernelGGL((k_reduce<MEAS_BLOCK>), dim3(1), dim3(MEAS_BLOCK), 0, c->stream, ws, nparts, sums_dev);
    HIP_TRY(hipGetLastError());
    c->timing.measure_launches++;
    return QCD_OK;
}

int qcd_measure_plaquette(qcd_ctx* c, double out[3])
{
    if (c && c->g.halo) return fail(QCD_ERR_STATE, "normalised measurement on a slab: add the *_sums of all contexts instead");
    double s[2];
    int rc = qcd_measure_plaquette_sums(c, s);
    if (rc) return rc;
    // SU(N)/suncpp/Measurements/Plq.h:90-92 with nd = 4
    const double v = (double)c->g.gsites;
    out[2] = (s[0] + s[1]) / (6 * v);
    out[0] = s[0] / (3 * v);
    out[1] = s[1] / (3 * v);
    return QCD_OK;
}

int qcd_measure_action(qcd_ctx* c, double out[3])
{
    double plq[3];
    int rc = qcd_measure_plaquette(c, plq);
    if (rc) return rc;
    // SU(N)/suncpp/Measurements/S.h:92-97
    for (int i = 0; i < 3; i++) out[i] = c->p.beta * (1 - plq[i] / c->p.group);
    return QCD_OK;
}

int qcd_measure_wilson(qcd_ctx* c, int R, int T, double* out)
{
    int rc = check_ctx(c, true, true, true);
    if (rc) return rc;
    if (!out) return fail(QCD_ERR_INVALID, "null output");
    if (R < 1 || T < 1) return fail(QCD_ERR_INVALID, "Wilson loop %d x %d", R, T);
    if (c->g.halo) return fail(QCD_ERR_UNSUPPORTED, "Wilson loop on a slab: lines of length R leave the one-slice halo");
    const int nb = cdiv(2L * c->g.vh_int, MEAS_BLOCK);
    if (nb > c->npartial) return fail(QCD_ERR_STATE, "partial buffer too small");
    Timed tm(c, 1);
    const dim3 grid(nb), block(MEAS_BLOCK);
    static const int rows_env = getenv("QCDGPU_AMD_WILSON_ROWS") ? atoi(getenv("QCDGPU_AMD_WILSON_ROWS")) : 1;   // diagnostic override: 0 = full 3 x 3 products
    if (c->p.group == 3 && !c->g.odd && rows_env) hipLaunchKernelGGL((k_wilson_rows<MEAS_BLOCK>), grid, block, 0, c->stream, c->U, c->g, R, T, c->partial);
    else if (c->p.group == 3) hipLaunchKernelGGL((k_wilson<3, MEAS_BLOCK>), grid, block, 0, c->stream, c->U, c->g, R, T, c->partial);
    else hipLaunchKernelGGL((k_wilson<2, MEAS_BLOCK>), grid, block, 0, c->stream, c->U, c->g, R, T, c->partial);
    if ((rc = tm.finish())) return rc;
    double s[2];
    rc = reduce_partials(c, nb, 1, s);
    if (rc) return rc;
    *out = s[0] / (3.0 * (double)c->g.gsites);   // SU(N)/suncl/suncl.cpp:1606-1611
    return QCD_OK;
}

int qcd_measure_fmunu_sums(qcd_ctx* c, int field, int index1, int index2, double sums[12])
{
    int rc = check_ctx(c, true, true, true);
    if (rc) return rc;
    if (field != 0 && field != 1) return fail(QCD_ERR_INVALID, "field %d is not 0 (xy, xz, yz) or 1 (xt, yt, zt)", field);
    const int top = c->p.group == 3 ? 8 : 3;   // Gell-Mann or Pauli matrices
    if (index1 < 1 || index1 > top || index2 < 1 || index2 > top)
        return fail(QCD_ERR_INVALID, "generator indices %d, %d outside 1..%d", index1, index2, top);
    if (!sums) return fail(QCD_ERR_INVALID, "null output");
    if (6L * c->npartial > c->partial_len) return fail(QCD_ERR_UNSUPPORTED, "reduction buffer too small");
    Timed tm(c, 1);
    int nparts = c->npartial;
    static const int march_env = getenv("QCDGPU_AMD_FMUNU_MARCH") ? atoi(getenv("QCDGPU_AMD_FMUNU_MARCH")) : 1;   // diagnostic override: 0 = one site per thread
    const int xseg = (c->p.group == 3 && !c->g.odd && march_env) ? march_segment(c->g) : 1;
    if (xseg > 1) {
        const Geom& g = c->g;
        nparts = cdiv((long)g.gslice * cdiv(g.nx, xseg), MEAS_BLOCK);
        if (nparts > c->npartial) return fail(QCD_ERR_STATE, "partial buffer too small");
        const dim3 grid(nparts), block(MEAS_BLOCK);
        if (xseg == 8 && field) hipLaunchKernelGGL((k_fmunu_march<MEAS_BLOCK, 8, 1>), grid, block, 0, c->stream, c->U, c->g, index1, index2, c->partial, nparts);
        else if (xseg == 8) hipLaunchKernelGGL((k_fmunu_march<MEAS_BLOCK, 8, 0>), grid, block, 0, c->stream, c->U, c->g, index1, index2, c->partial, nparts);
        else if (field) hipLaunchKernelGGL((k_fmunu_march<MEAS_BLOCK, 4, 1>), grid, block, 0, c->stream, c->U, c->g, index1, index2, c->partial, nparts);
        else hipLaunchKernelGGL((k_fmunu_march<MEAS_BLOCK, 4, 0>), grid, block, 0, c->stream, c->U, c->g, index1, index2, c->partial, nparts);
    } else if (c->p.group == 3)
        hipLaunchKernelGGL((k_fmunu<3, MEAS_BLOCK>), dim3(c->npartial), dim3(MEAS_BLOCK), 0, c->stream, c->U, c->g, field, index1,
                           index2, c->partial, c->npartial);
    else
        hipLaunchKernelGGL((k_fmunu<2, MEAS_BLOCK>), dim3(c->npartial), dim3(MEAS_BLOCK), 0, c->stream, c->U, c->g, field, index1,
                           index2, c->partial, c->npartial);
    HIP_TRY(hipGetLastError());
    rc = reduce_partials(c, nparts, 6, sums);
    if (rc) return rc;
    return tm.finish();
}

int qcd_measure_fmunu(qcd_ctx* c, int field, int index1, int index2, double out[12])
{
    int rc = qcd_measure_fmunu_sums(c, field, index1, index2, out);
    if (rc) return rc;
    const double sites = 2.0 * c->g.vh_int;
    for (int i = 0; i < 12; i++) out[i] /= sites;
    return QCD_OK;
}

// the loop of every owned space site into c->loops (and the four global sums into c->partial)
// *nb: the number of partial sums per output it left
static int launch_polyakov(qcd_ctx* c, int* nb)
{
    const int n3 = c->g.nx * c->g.ny * c->g.nz;
    static const int split_env = getenv("QCDGPU_AMD_POLYAKOV_SPLIT") ? atoi(getenv("QCDGPU_AMD_POLYAKOV_SPLIT")) : 1;   // diagnostic override: 0 = one thread per line
    const int nbs = cdiv(n3, MEAS_BLOCK / 4);
    if (split_env && c->g.nt >= 8 && nbs <= c->npartial) {
        const dim3 grid(nbs), block(MEAS_BLOCK);
        if (c->p.group == 3) hipLaunchKernelGGL((k_polyakov_split<3, MEAS_BLOCK>), grid, block, 0, c->stream, c->U, c->g, c->partial, nbs, c->loops);
        else hipLaunchKernelGGL((k_polyakov_split<2, MEAS_BLOCK>), grid, block, 0, c->stream, c->U, c->g, c->partial, nbs, c->loops);
        HIP_TRY(hipGetLastError());
        *nb = nbs;
        return QCD_OK;
    }
    *nb = cdiv(n3, MEAS_BLOCK);
    const dim3 grid(*nb), block(MEAS_BLOCK);
    if (c->p.group == 3) hipLaunchKernelGGL((k_polyakov<3, MEAS_BLOCK>), grid, block, 0, c->stream, c->U, c->g, c->partial, *nb, c->loops);
    else hipLaunchKernelGGL((k_polyakov<2, MEAS_BLOCK>), grid, block, 0, c->stream, c->U, c->g, c->partial, *nb, c->loops);
    HIP_TRY(hipGetLastError());
    return QCD_OK;
}

// shared driver of the two profile measurements.  A profile reads the per-site values that the
// last Polyakov (plaquette) measurement left, when no other call came between: a sample's global
// value and its three profiles then cost one pass over the links instead of four.
static int measure_profile(qcd_ctx* c, int axis, double* sums, bool polyakov)
{
    int rc = check_ctx(c, true, true, true);
    if (rc) return rc;
    if (axis < 0 || axis > 2) return fail(QCD_ERR_INVALID, "axis %d is not 0 (x), 1 (y) or 2 (z)", axis);
    if (!sums) return fail(QCD_ERR_INVALID, "null output");
    const Geom& g = c->g;
    const int ext = axis == 0 ? g.nx : (axis == 1 ? g.ny : g.nz);
    if (polyakov) {
        const int n3 = g.nx * g.ny * g.nz;
        const int nb = cdiv(n3, MEAS_BLOCK);
        if (nb > c->npartial || ext > c->partial_len || 2 * ext > c->result_len)
            return fail(QCD_ERR_UNSUPPORTED, "profile of %d planes does not fit the reduction buffers", ext);
        Timed tm(c, 1);
        int nb_used = 0;
        if (!c->loops_fresh && (rc = launch_polyakov(c, &nb_used))) return rc;
        // one workgroup per plane: its pair is the plane's result and goes straight to the host's result words
        hipLaunchKernelGGL((k_loops_profile<MEAS_BLOCK>), dim3(ext), dim3(MEAS_BLOCK), 0, c->stream, c->loops, g, axis,
                           (double2*)c->result_host_dev);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (int i = 0; i < 2 * ext; i++) sums[i] = c->result_host[i];
        c->loops_fresh = true;
        return tm.finish();
    }
    const long sites = (long)g.nx * g.gslice;
    const int plane = (int)(sites / ext);
    int chunks = cdiv(plane, MEAS_BLOCK);
    if (chunks > 64) chunks = 64;
    if ((long)ext * chunks > c->partial_len || 2 * ext > c->result_len)
        return fail(QCD_ERR_UNSUPPORTED, "profile of %d planes does not fit the reduction buffers", ext);
    bool have = c->sites_fresh;
    if (!c->site_vals) {   // first plaquette profile of this context
        hipError_t e = hipMalloc(&c->site_vals, (size_t)2 * g.vh_int * sizeof(double2));
        if (e != hipSuccess) {
            c->site_vals = nullptr;
            return fail(e == hipErrorOutOfMemory ? QCD_ERR_NOMEM : QCD_ERR_HIP, "hipMalloc(per-site plaquettes): %s", hipGetErrorString(e));
        }
        have = false;
    }
    Timed tm(c, 1);
    int nparts_unused = 0;
    if (!have && (rc = launch_plaquette(c, &nparts_unused))) return rc;
    hipLaunchKernelGGL((k_sites_profile<MEAS_BLOCK>), dim3(ext * chunks), dim3(MEAS_BLOCK), 0, c->stream, c->site_vals, g, axis, chunks,
                       plane, c->partial);
    HIP_TRY(hipGetLastError());
    if ((rc = reduce_partials(c, chunks, ext, sums))) return rc;
    c->sites_fresh = true;
    return tm.finish();
}

int qcd_measure_plaquette_profile_sums(qcd_ctx* c, int axis, double* sums) { return measure_profile(c, axis, sums, false); }
int qcd_measure_polyakov_profile_sums(qcd_ctx* c, int axis, double* sums) { return measure_profile(c, axis, sums, true); }

int qcd_measure_polyakov_sums(qcd_ctx* c, double sums[4])
{
    int rc = check_ctx(c, true, true, true);
    if (rc) return rc;
    if (!sums) return fail(QCD_ERR_INVALID, "null output");
    const int n3 = c->g.nx * c->g.ny * c->g.nz;
    int nb = cdiv(n3, MEAS_BLOCK);
    if (nb > c->npartial) return fail(QCD_ERR_STATE, "partial buffer too small");
    Timed tm(c, 1);
    if ((rc = launch_polyakov(c, &nb))) return rc;
    if ((rc = tm.finish())) return rc;
    if ((rc = reduce_partials(c, nb, 2, sums))) return rc;
    c->loops_fresh = true;
    return QCD_OK;
}

int qcd_measure_polyakov(qcd_ctx* c, double out[4])
{
    if (c && c->g.halo) return fail(QCD_ERR_STATE, "normalised measurement on a slab: add the *_sums of all contexts instead");
    double s[4];
    int rc = qcd_measure_polyakov_sums(c, s);
    if (rc) return rc;
    // SU(N)/suncl/su3cpu.cpp:1427-1430
    const double n3 = (double)c->p.L[0] * c->p.L[1] * c->p.L[2];
    const double n = c->p.group;
    out[0] = s[0] / (n3 * n);
    out[1] = s[1] / (n3 * n);
    out[2] = s[2] / (n3 * n * n);
    out[3] = s[3] / (n3 * n * n * n * n);
    return QCD_OK;
}

static int transfer(qcd_ctx* c, void* host, size_t bytes, int rowsize, bool up)
{
    int rc = check_ctx(c, !up);
    if (rc) return rc;
    const Geom& g = c->g;
    const int rows = (c->p.group == 3) ? 12 : 4;
    const long nsite = (long)g.nx * g.gslice;
    if (!host) return fail(QCD_ERR_INVALID, "null host table");
    if (rowsize < nsite) return fail(QCD_ERR_INVALID, "rowsize %d < owned sites %ld", rowsize, nsite);
    const size_t need = (size_t)rows * rowsize * 4 * sizeof(real);
    if (bytes != need) return fail(QCD_ERR_INVALID, "table is %zu bytes, expected rows*rowsize*%d = %zu", bytes, (int)(4 * sizeof(real)), need);
    real4* dev = nullptr;
    HIP_TRY(hipMalloc(&dev, need));
    hipError_t e = hipSuccess;
    if (up) {
        e = hipMemcpyAsync(dev, host, need, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) rc = convert_slices(c, dev, rowsize, g.halo, g.nx, true);
    } else {
        e = hipMemsetAsync(dev, 0, need, c->stream);
        if (e == hipSuccess) rc = convert_slices(c, dev, rowsize, g.halo, g.nx, false);
        if (e == hipSuccess && rc == QCD_OK) e = hipMemcpyAsync(host, dev, need, hipMemcpyDeviceToHost, c->stream);
    }
    hipError_t e2 = hipStreamSynchronize(c->stream);
    (void)hipFree(dev);
    if (e != hipSuccess) return fail(QCD_ERR_HIP, "copy: %s", hipGetErrorString(e));
    if (e2 != hipSuccess) return fail(QCD_ERR_HIP, "sync: %s", hipGetErrorString(e2));
    if (rc) return rc;
    if (up && !c->initialized) {
        // a configuration supplied by the caller: streams start from their seeds
        if (c->p.start_mode == QCD_START_GID) return fail(QCD_ERR_STATE, "upload before qcd_init_lattice in gid mode");
        rc = seed_streams(c);
        if (rc) return rc;
        c->initialized = true;
    }
    return QCD_OK;
}

int qcd_upload(qcd_ctx* c, const void* host_table, size_t bytes, int rowsize)
{
    return transfer(c, const_cast<void*>(host_table), bytes, rowsize, true);
}

int qcd_download(qcd_ctx* c, void* host_table, size_t bytes, int rowsize)
{
    return transfer(c, host_table, bytes, rowsize, false);
}

// State of this context's RANLUX streams, for checkpoints: [draws u64][ctl ns][slot0 25*ns]
// [slot1 25*ns] (32-bit words; the two slots in word-major order [w][stream]).  The reference
// restores its generator by replaying produce() PRNG_counter times (SU(N)/suncl/suncl.cpp:5349-5351);
// per-thread streams are saved instead.
size_t qcd_rng_state_bytes(const qcd_ctx* c)
{
    if (!c) return 0;
    return sizeof(unsigned long long) + (size_t)(2 * RLX_STATE_WORDS + 1) * (size_t)c->rv.ns * sizeof(unsigned);
}

static int rng_state_copy(qcd_ctx* c, void* host, size_t bytes, bool up)
{
    int rc = check_ctx(c, true);
    if (rc) return rc;
    if (!host) return fail(QCD_ERR_INVALID, "null buffer");
    if (bytes != qcd_rng_state_bytes(c)) return fail(QCD_ERR_INVALID, "generator state is %zu bytes, got %zu", qcd_rng_state_bytes(c), bytes);
    // the caller's format has the 24 words and the carry of a state one per 32-bit word; the device
    // keeps them packed (rlx_pack4): converted here
    const size_t ns = (size_t)c->rv.ns;
    char* h = (char*)host;
    const hipMemcpyKind kind = up ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost;
    auto cp = [&](void* dev, void* hp, size_t n) -> hipError_t {
        return up ? hipMemcpyAsync(dev, hp, n, kind, c->stream) : hipMemcpyAsync(hp, dev, n, kind, c->stream);
    };
    if (up) {   // the saved total goes into the first partial count
        HIP_TRY(hipMemsetAsync(c->draws_dev, 0, DRAWS_BYTES, c->stream));
        HIP_TRY(hipMemcpyAsync(c->draws_dev, h, sizeof(unsigned long long), hipMemcpyHostToDevice, c->stream));
    } else {
        unsigned long long total = 0;
        if ((rc = read_draws(c, &total))) return rc;
        memcpy(h, &total, sizeof(total));
    }
    h += sizeof(unsigned long long);
    // control words: which of a stream's two slots holds its state in front of the next batch.  The
    // device has no such words (all streams of an x-slice are in the same slot, slice_slot): written
    // from that on the way out; on the way in each stream's current state goes to slot 0
    unsigned* ctlw = (unsigned*)h;
    if (!up)
        for (size_t i = 0; i < ns; i++) ctlw[i] = c->slice_slot[i / (size_t)(c->g.slice_h / c->g.share)];
    h += ns * sizeof(unsigned);
    std::vector<unsigned> packed((size_t)2 * RLX_SLOT_WORDS * ns);   // both slots, as on the device
    unsigned* wide = (unsigned*)h;                                   // [slot][w][stream], w in [0, 25)
    auto convert = [&](bool pack) {
        for (int slot = 0; slot < 2; slot++) {
            unsigned* wd = wide + (size_t)slot * RLX_STATE_WORDS * ns;
            for (size_t i = 0; i < ns; i++) {
                const int dev_slot = pack ? (slot ^ (int)(ctlw[i] & 1u)) : slot;
                unsigned* pk = packed.data() + (size_t)dev_slot * RLX_SLOT_WORDS * ns;
                RlxRegs r;
                unsigned d[RLX_SLOT_WORDS];
                if (pack) {
                    for (int w = 0; w < 24; w++) r.s[w] = wd[(size_t)w * ns + i] & 0xffffffu;
                    r.c = wd[(size_t)24 * ns + i];
                    rlx_pack(r, d);
                    for (int k = 0; k < RLX_SLOT_WORDS; k++) pk[rlx_slot_index(k, ns, i)] = d[k];
                } else {
                    for (int k = 0; k < RLX_SLOT_WORDS; k++) d[k] = pk[rlx_slot_index(k, ns, i)];
                    rlx_unpack(d, r);
                    for (int w = 0; w < 24; w++) wd[(size_t)w * ns + i] = r.s[w];
                    wd[(size_t)24 * ns + i] = r.c;
                }
            }
        }
    };
    if (up) {
        convert(true);
        memset(c->slice_slot, 0, (size_t)c->g.nx);
    }
    HIP_TRY(cp(c->rv.slot0, packed.data(), packed.size() * sizeof(unsigned)));   // slot1 follows slot0 in one allocation
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (!up) convert(false);
    return QCD_OK;
}

int qcd_rng_state_download(qcd_ctx* c, void* host, size_t bytes) { return rng_state_copy(c, host, bytes, false); }
int qcd_rng_state_upload(qcd_ctx* c, const void* host, size_t bytes) { return rng_state_copy(c, const_cast<void*>(host), bytes, true); }

size_t qcd_halo_bytes(const qcd_ctx* c)
{
    if (!c) return 0;
    return (size_t)c->ncomp * c->g.slice_h * sizeof(real2);
}

static int halo_op(qcd_ctx* c, int side, int dir, int parity, void* buf, bool pack)
{
    int rc = check_ctx(c, true, false);
    if (rc) return rc;
    const hipStream_t stream = c->stream;
    if (!c->g.halo) return fail(QCD_ERR_STATE, "context has no halo slices");
    if (side != 0 && side != 1) return fail(QCD_ERR_INVALID, "side %d", side);
    if (dir < 0 || dir > 3) return fail(QCD_ERR_INVALID, "direction %d", dir);
    if (parity != 0 && parity != 1) return fail(QCD_ERR_INVALID, "parity %d", parity);
    if (!buf) return fail(QCD_ERR_INVALID, "null buffer");
    // stored slices: 0 = left halo, 1..nx = owned, nx+1 = right halo
    const int slice = pack ? (side == 0 ? 1 : c->g.nx) : (side == 0 ? 0 : c->g.nx + 1);
    Timed tm(c, 2);
    const dim3 grid(cdiv((long)c->ncomp * c->g.slice_h, MEAS_BLOCK)), block(MEAS_BLOCK);
    if (c->p.group == 3) {
        if (pack) hipLaunchKernelGGL((k_halo<3, true, MEAS_BLOCK>), grid, block, 0, stream, c->U, c->g, dir, parity, slice, (real2*)buf);
        else hipLaunchKernelGGL((k_halo<3, false, MEAS_BLOCK>), grid, block, 0, stream, c->U, c->g, dir, parity, slice, (real2*)buf);
    } else {
        if (pack) hipLaunchKernelGGL((k_halo<2, true, MEAS_BLOCK>), grid, block, 0, stream, c->U, c->g, dir, parity, slice, (real2*)buf);
        else hipLaunchKernelGGL((k_halo<2, false, MEAS_BLOCK>), grid, block, 0, stream, c->U, c->g, dir, parity, slice, (real2*)buf);
    }
    return tm.finish();
}

int qcd_halo_pack(qcd_ctx* c, int side, int dir, int parity, void* dev_buf)
{
    return halo_op(c, side, dir, parity, dev_buf, true);
}

int qcd_halo_unpack(qcd_ctx* c, int side, int dir, int parity, const void* dev_buf)
{
    return halo_op(c, side, dir, parity, const_cast<void*>(dev_buf), false);
}

int qcd_halo_region(qcd_ctx* c, int side, int owned, int dir, int parity, void** first, size_t* piece_bytes, size_t* pitch_bytes, int* pieces)
{
    if (!c || !first || !piece_bytes || !pitch_bytes || !pieces) return fail(QCD_ERR_INVALID, "null argument");
    if (!c->g.halo) return fail(QCD_ERR_STATE, "context has no halo slices");
    if (side != 0 && side != 1) return fail(QCD_ERR_INVALID, "side %d", side);
    if (dir < 0 || dir > 3) return fail(QCD_ERR_INVALID, "direction %d", dir);
    if (parity != 0 && parity != 1) return fail(QCD_ERR_INVALID, "parity %d", parity);
    // stored slices: 0 = left halo, 1..nx = owned, nx+1 = right halo; a slice of one component
    // array is one run of slice_h elements (x is the slowest index of a site's half index)
    const int slice = owned ? (side == 0 ? 1 : c->g.nx) : (side == 0 ? 0 : c->g.nx + 1);
    const size_t base = c->p.group == 3 ? comp_base<3>(c->g, dir, parity) : comp_base<2>(c->g, dir, parity);
    *first = (void*)(c->U + base + (size_t)slice * c->g.slice_h);
    *piece_bytes = (size_t)c->g.slice_h * sizeof(real2);
    *pitch_bytes = (size_t)c->g.vh_alloc * sizeof(real2);
    *pieces = c->ncomp;
    return QCD_OK;
}

static int edge_buffers(qcd_ctx* c)
{
    if (c->edge_buf) return QCD_OK;
    const size_t bytes = qcd_halo_bytes(c);
    HIP_TRY(hipMalloc(&c->edge_buf, 2 * bytes));  // [0, bytes): outgoing, [bytes, 2*bytes): incoming
    HIP_TRY(hipEventCreateWithFlags(&c->edge_ev, hipEventDisableTiming));
    return QCD_OK;
}

int qcd_halo_copy(qcd_ctx* src, int pack_side, qcd_ctx* dst, int recv_side, int dir, int parity)
{
    if (!src || !dst) return fail(QCD_ERR_INVALID, "null context");
    if (qcd_halo_bytes(src) != qcd_halo_bytes(dst) || src->p.group != dst->p.group)
        return fail(QCD_ERR_INVALID, "parts with different slice shapes");
    const size_t bytes = qcd_halo_bytes(src);
    int rc = check_ctx(src, true);
    if (rc) return rc;
    if ((rc = edge_buffers(src))) return rc;
    if ((rc = qcd_halo_pack(src, pack_side, dir, parity, src->edge_buf))) return rc;
    HIP_TRY(hipEventRecord(src->edge_ev, src->stream));

    if ((rc = check_ctx(dst, true))) return rc;
    if ((rc = edge_buffers(dst))) return rc;
    HIP_TRY(hipStreamWaitEvent(dst->stream, src->edge_ev, 0));
    const real2* from = src->edge_buf;
    if (src->p.device != dst->p.device) {
        real2* in = dst->edge_buf + bytes / sizeof(real2);
        HIP_TRY(hipMemcpyPeerAsync(in, dst->p.device, src->edge_buf, src->p.device, bytes, dst->stream));
        from = in;
    }
    if ((rc = qcd_halo_unpack(dst, recv_side, dir, parity, from))) return rc;
    // the sender must not repack its outgoing buffer before the receiver has consumed it
    HIP_TRY(hipEventRecord(dst->edge_ev, dst->stream));
    HIP_TRY(hipSetDevice(src->p.device));
    HIP_TRY(hipStreamWaitEvent(src->stream, dst->edge_ev, 0));
    return QCD_OK;
}

int qcd_device_count(int* count)
{
    if (!count) return fail(QCD_ERR_INVALID, "null output");
    HIP_TRY(hipGetDeviceCount(count));
    return QCD_OK;
}

int qcd_device_name(int device, char* buf, size_t buflen)
{
    if (!buf || buflen == 0) return fail(QCD_ERR_INVALID, "null output");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    snprintf(buf, buflen, "%s (%s)", prop.name, prop.gcnArchName);
    return QCD_OK;
}

int qcd_set_profiling(qcd_ctx* c, int on)
{
    int rc = check_ctx(c, false);
    if (rc) return rc;
    c->profiling = on != 0;
    return QCD_OK;
}

int qcd_get_timing(qcd_ctx* c, qcd_timing* out)
{
    if (!c || !out) return fail(QCD_ERR_INVALID, "null argument");
    *out = c->timing;
    return QCD_OK;
}

namespace {
int timed_sweeps(qcd_ctx* c, int n, double* elapsed_ms, unsigned long long* launches, bool overrelax);
}
int qcd_timed_sweeps(qcd_ctx* c, int n, double* elapsed_ms, unsigned long long* launches)
{
    return timed_sweeps(c, n, elapsed_ms, launches, false);
}
int qcd_timed_overrelax_sweeps(qcd_ctx* c, int n, double* elapsed_ms, unsigned long long* launches)
{
    return timed_sweeps(c, n, elapsed_ms, launches, true);
}
namespace {
int timed_sweeps(qcd_ctx* c, int n, double* elapsed_ms, unsigned long long* launches, bool overrelax)
{
    int rc = check_ctx(c, true);
    if (rc) return rc;
    if (n < 1 || !elapsed_ms) return fail(QCD_ERR_INVALID, "bad arguments");
    if (c->profiling) return fail(QCD_ERR_STATE, "per-launch profiling is on: switch it off for a bracketed timing");
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    const unsigned long long before = c->timing.update_launches;
    HIP_TRY(hipEventRecord(a, c->stream));
    rc = overrelax ? qcd_overrelax_sweep(c, n) : qcd_sweep(c, n);
    hipError_t e = hipEventRecord(b, c->stream);
    if (e == hipSuccess) e = hipEventSynchronize(b);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    if (rc) return rc;
    if (e != hipSuccess) return fail(QCD_ERR_HIP, "event timing: %s", hipGetErrorString(e));
    *elapsed_ms = ms;
    if (launches) *launches = c->timing.update_launches - before;
    return QCD_OK;
}
}  // namespace

int qcd_rng_draws(qcd_ctx* c, unsigned long long* draws)
{
    int rc = check_ctx(c, true);
    if (rc) return rc;
    if (!draws) return fail(QCD_ERR_INVALID, "null output");
    return read_draws(c, draws);
}

// ---- host-only helpers exported for the CPU test-suite (no device needed) -------------------

// Generator self-check, the counterpart of PRNG::check (SU(N)/random/random.cpp:339-402): the
// first `samples` numbers of streams 0 .. streams-1 produced on the device are compared, bit for
// bit, with the host generator's.  *errors = number of values that differ.
int qcd_prng_check(int device, unsigned rand_series, int lux_level, int streams, int samples, unsigned* errors)
{
    if (!errors || streams < 1 || samples < 1 || lux_level < 0 || lux_level > 4 || rand_series == 0)
        return fail(QCD_ERR_INVALID, "bad arguments");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(QCD_ERR_INVALID, "device %d of %d", device, ndev);
    HIP_TRY(hipSetDevice(device));
    const int nskip = rlx_nskip_for_level(lux_level);
    const int batches = (samples + 23) / 24;
    const int ns = streams;
    std::vector<int> seeds = stream_seeds(rand_series, 0, ns);
    std::vector<unsigned> st((size_t)RLX_STATE_WORDS * ns);
    std::vector<RlxRegs> regs((size_t)ns);
    for (int i = 0; i < ns; i++) {
        rlx_seed_state(seeds[(size_t)i], nskip, regs[(size_t)i]);
        for (int w = 0; w < 24; w++) st[(size_t)w * ns + i] = regs[(size_t)i].s[w];
        st[(size_t)24 * ns + i] = regs[(size_t)i].c;
    }
    unsigned* d_state = nullptr;
    float* d_out = nullptr;
    const size_t nout = (size_t)24 * batches * ns;
    HIP_TRY(hipMalloc(&d_state, st.size() * sizeof(unsigned)));
    hipError_t e = hipMalloc(&d_out, nout * sizeof(float));
    if (e != hipSuccess) {
        (void)hipFree(d_state);
        return fail(QCD_ERR_NOMEM, "hipMalloc: %s", hipGetErrorString(e));
    }
    std::vector<float> got(nout);
    e = hipMemcpy(d_state, st.data(), st.size() * sizeof(unsigned), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL((k_rlx_dump<MEAS_BLOCK>), dim3(cdiv((long)ns, MEAS_BLOCK)), dim3(MEAS_BLOCK), 0, 0, d_state, ns,
                           nskip / 24, nskip % 24, batches, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(got.data(), d_out, nout * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d_state);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(QCD_ERR_HIP, "generator check: %s", hipGetErrorString(e));
    unsigned bad = 0;
    for (int i = 0; i < ns; i++) {
        RlxRegs r = regs[(size_t)i];
        float batch[24];
        for (int b = 0; b < batches; b++) {
            rlx_batch(r, batch, 1);
            rlx_skip(r, nskip / 24, nskip % 24);
            for (int k = 0; k < 24 && 24 * b + k < samples; k++) {
                const float d = got[((size_t)24 * b + k) * ns + i];
                if (memcmp(&d, &batch[k], sizeof(float)) != 0) bad++;
            }
        }
    }
    *errors = bad;
    return QCD_OK;
}

// Self-check of the heat-bath's own fp64 reciprocal / square root / logarithm (fastmath.h)
// against the host libm: n arguments x_i in (0, 1) (as floats k * 2^-24, the form the heat-bath
// feeds them), worst error in units of the last place of: log(1 - x), sqrt(x), 1/(x + 1e-3),
// 1/sqrt(x + 1e-3).
int qcd_math_check(int device, int n, double max_ulp[4])
{
    if (!max_ulp || n < 1) return fail(QCD_ERR_INVALID, "bad arguments");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(QCD_ERR_INVALID, "device %d of %d", device, ndev);
    HIP_TRY(hipSetDevice(device));
    std::vector<double> in((size_t)n), out((size_t)4 * n);
    unsigned long long lcg = 88172645463325252ull;
    for (int i = 0; i < n; i++) {
        lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
        unsigned k = (unsigned)(lcg >> 40);                 // 24 bits
        if (i % 7 == 0) k >>= (i / 7) % 24;                 // also small arguments (1 - x close to 1)
        if (i % 11 == 0) k = 0xFFFFFFu - (k >> ((i / 11) % 24));   // and arguments close to 1
        if (k == 0) k = 1;
        in[(size_t)i] = (double)((float)k * 5.9604644775390625e-8f);
    }
    double *d_in = nullptr, *d_out = nullptr;
    HIP_TRY(hipMalloc(&d_in, in.size() * sizeof(double)));
    hipError_t e = hipMalloc(&d_out, out.size() * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(d_in, in.data(), in.size() * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL((k_math_check<MEAS_BLOCK>), dim3(cdiv((long)n, MEAS_BLOCK)), dim3(MEAS_BLOCK), 0, 0, d_in, n, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out.data(), d_out, out.size() * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(QCD_ERR_HIP, "math check: %s", hipGetErrorString(e));
    auto ulps = [](double got, long double want) {
        if (want == 0.0L) return got == 0.0 ? 0.0 : 1e30;
        const double u = std::ldexp(1.0, std::ilogb((double)want) - 52);
        return (double)(fabsl((long double)got - want) / u);
    };
    for (int k = 0; k < 4; k++) max_ulp[k] = 0.0;
    for (int i = 0; i < n; i++) {
        const double x = in[(size_t)i];
        const long double w[4] = {logl(1.0L - (long double)(float)x), sqrtl((long double)x), 1.0L / ((long double)x + 1e-3L),
                                  1.0L / sqrtl((long double)(x + 1e-3))};
        const long double arg2 = (long double)(x + 1e-3);   // the kernel adds in fp64
        const long double w2 = 1.0L / arg2;
        for (int k = 0; k < 4; k++) {
            const long double want = (k == 2) ? w2 : w[k];
            const double u = ulps(out[(size_t)k * n + i], want);
            if (u > max_ulp[k]) max_ulp[k] = u;
        }
    }
    return QCD_OK;
}

// Self-check of the accept step's fp64 cosine and sine of 2 pi r (MathFast::sincos_2pi: table + Taylor series) against
// the host libm in long double: n arguments r in [0, 1) of the forms the generator delivers -- k 2^-24, and below 2^-12
// with the bits of its correction term behind them; worst absolute errors.
int qcd_trig_check(int device, int n, double max_abs_err[2])
{
    if (!max_abs_err || n < 1) return fail(QCD_ERR_INVALID, "bad arguments");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(QCD_ERR_INVALID, "device %d of %d", device, ndev);
    HIP_TRY(hipSetDevice(device));
    std::vector<float> in((size_t)n);
    std::vector<double> out((size_t)2 * n);
    unsigned long long lcg = 88172645463325252ull;
    for (int i = 0; i < n; i++) {
        lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
        unsigned k = (unsigned)(lcg >> 40);                 // 24 bits
        float r = (float)k * 5.9604644775390625e-8f;
        if (i % 5 == 0) {
            // a value below 2^-12 with the correction term (ranlux.h, rlx_batch)
            k &= 4095u;
            const unsigned sj = (unsigned)(lcg >> 16) & 0xFFFFFFu;
            r = (float)k * 5.9604644775390625e-8f;
            r = r + 5.9604644775390625e-8f * ((float)sj * 5.9604644775390625e-8f);
        }
        if (i % 9 == 0) r = (float)((k & 0xFF0000u) | (i % 18 == 0 ? 0x7FFFu : 0x8000u)) * 5.9604644775390625e-8f;   // either side of the table's midpoints
        if (i < 256) r = (float)((unsigned)i << 16) * 5.9604644775390625e-8f;   // the table's own angles
        in[(size_t)i] = r;
    }
    float* d_in = nullptr;
    double* d_out = nullptr;
    HIP_TRY(hipMalloc(&d_in, in.size() * sizeof(float)));
    hipError_t e = hipMalloc(&d_out, out.size() * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(d_in, in.data(), in.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL((k_trig_check<MEAS_BLOCK>), dim3(cdiv((long)n, MEAS_BLOCK)), dim3(MEAS_BLOCK), 0, 0, d_in, n, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out.data(), d_out, out.size() * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(QCD_ERR_HIP, "trig check: %s", hipGetErrorString(e));
    max_abs_err[0] = max_abs_err[1] = 0.0;
    const long double two_pi = 2.0L * acosl(-1.0L);
    for (int i = 0; i < n; i++) {
        const long double a = two_pi * (long double)in[(size_t)i];
        const double ec = (double)fabsl((long double)out[(size_t)i] - cosl(a));
        const double es = (double)fabsl((long double)out[(size_t)n + i] - sinl(a));
        if (ec > max_abs_err[0]) max_abs_err[0] = ec;
        if (es > max_abs_err[1]) max_abs_err[1] = es;
    }
    return QCD_OK;
}

// n values of stream `stream` produced by the batch formulation the kernels use
int qcd_host_ranlux_stream(unsigned rand_series, int stream, int lux_level, int n, float* out)
{
    if (!out || n < 0 || stream < 0) return fail(QCD_ERR_INVALID, "bad arguments");
    const int nskip = rlx_nskip_for_level(lux_level);
    std::vector<int> seeds = stream_seeds(rand_series, stream, 1);
    RlxRegs r;
    rlx_seed_state(seeds[0], nskip, r);
    float batch[24];
    int produced = 0;
    while (produced < n) {
        rlx_batch(r, batch, 1);
        for (int k = 0; k < 24 && produced < n; k++) out[produced++] = batch[k];
        rlx_skip(r, nskip / 24, nskip % 24);
    }
    return QCD_OK;
}

// the closed-form gid tables, as uploaded by qcd_init_lattice
int qcd_host_gid_link(int group, int gid, int dir, int sites, double quads[12])
{
    if (!quads) return fail(QCD_ERR_INVALID, "null output");
    gid_link(group, gid, dir, sites, quads);
    return QCD_OK;
}

int qcd_host_gid_randoms(int gid, int nhit, int sites, float out[6])
{
    if (!out) return fail(QCD_ERR_INVALID, "null output");
    gid_randoms(gid, nhit, sites, out);
    return QCD_OK;
}

// packs and unpacks n pseudo-random generator states (every word below 2^24, carry 0 or 1) the way the
// device keeps them; *errors = states that did not survive, or whose packed form is not 19 words
int qcd_host_rlx_pack_check(unsigned seed, int n, unsigned* errors)
{
    if (!errors) return fail(QCD_ERR_INVALID, "null output");
    unsigned x = seed | 1u, bad = 0;
    auto next = [&x]() { x = x * 1664525u + 1013904223u; return x >> 8; };   // 24 bits
    for (int i = 0; i < n; i++) {
        RlxRegs r, t;
        for (int w = 0; w < 24; w++) r.s[w] = (i == 0) ? 0xffffffu : (i == 1 ? 0u : next());
        r.c = next() & 1u;
        unsigned d[RLX_SLOT_WORDS];
        rlx_pack(r, d);
        rlx_unpack(d, t);
        bool ok = t.c == r.c && d[RLX_PACKED_WORDS] == 0u;
        for (int w = 0; w < 24; w++) ok = ok && t.s[w] == r.s[w];
        if (!ok) bad++;
    }
    *errors = bad;
    return QCD_OK;
}

}  // extern "C"

#ifdef QCD_GH_STATS
// diagnostic build only: sleeps the two sides of the gather / heat-bath pairs spent waiting for each other
extern "C" int qcd_debug_gh_spins(unsigned long long out[2], int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(QCD_NS::g_gh_spins), 16);
    if (e == hipSuccess && reset) {
        unsigned long long z[2] = {0, 0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(QCD_NS::g_gh_spins), z, 16);
    }
    return (int)e;
}
#endif

#ifdef QCD_TRACE
// diagnostic build only: copies the per-wave phase stamps of the last k_update launch
extern "C" int qcd_debug_read_trace(void* dst, size_t bytes)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(QCD_NS::g_trace), bytes);
}
#endif
