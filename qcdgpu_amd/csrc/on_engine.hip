// on_engine.hip -- C-ABI of the O(N) scalar path (qcd_on_* in include/qcdgpu_amd.h) over on_kernels.h.
//
// A context owns the whole lattice on one device: the field (float or double, split by parity),
// one RANLUX stream per thread of an update pass, counters and reduction buffers.  Compiled with
// floating-point contraction off, so that the mixed float/double expressions of the action round
// exactly where the kernel source text says they do.
#include "../../include/qcdgpu_amd.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "host_seed.h"
#include "on_kernels.h"

using namespace QCD_NS;

namespace QCD_NS {
int fail_msg(int code, const char* msg);
}

namespace {

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return fail_msg(code, buf);
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(QCD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                               \
    } while (0)

constexpr int ON_BLOCK = 256;
constexpr int ON_MEAS_BLOCK = 256;

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace

struct qcd_on_ctx {
    qcd_on_params p;
    OnGeom g;
    double sqrt_z_lambda_zeta, max_U;
    hipStream_t stream;
    void* phi;                     // F[2][vh]
    size_t elem;                   // sizeof(F)
    RlxView rv;
    unsigned long long* counters;  // [0] numbers drawn, [1] ACC_RATE sum of the last even pass, [2] odd
    unsigned long long* wg_counts; // per workgroup of an update pass: numbers drawn, ACC_RATE part
    double* partial;
    int npartial;
    double* result_dev;
    double* result_host;           // pinned, 8 doubles
    hipEvent_t ev0, ev1;
    bool initialized;
};

namespace {

template <typename F>
OnPar<F> params_of(const qcd_on_ctx* c)
{
    OnPar<F> q;
    q.z = (F)c->p.z;
    q.zeta = (F)c->p.zeta;
    q.eta = (F)c->p.eta;
    q.b = (F)c->p.b;
    q.sqrt_z_lambda_zeta = (F)c->sqrt_z_lambda_zeta;
    q.max_U = (F)c->max_U;
    q.skgroup = c->p.sk_group;
    return q;
}

int check(const qcd_on_ctx* c, bool need_init)
{
    if (!c) return fail(QCD_ERR_INVALID, "null context");
    if (need_init && !c->initialized) return fail(QCD_ERR_STATE, "field not initialised: call qcd_on_init_lattice or qcd_on_upload first");
    HIP_TRY(hipSetDevice(c->p.device));
    return QCD_OK;
}

// stream i is seeded like instance i of the reference's device generator (host_seed.h); the
// luxury skip in front of the first batch is spread over host threads (8.4 M streams at 64^4)
int seed_streams(qcd_on_ctx* c)
{
    const int ns = c->g.vh;
    const int nskip = rlx_nskip_for_level(c->p.lux_level);
    const std::vector<int> seeds = stream_seeds(c->p.rand_series, 0, ns);
    const int ns_pad = cdiv(ns, ON_RLX_TILE) * ON_RLX_TILE;
    std::vector<unsigned> st((size_t)RLX_SLOT_WORDS * ns_pad);
    auto at = [](int w, int i) {   // [tile][quad][lane][4], see RlxTileHome
        return (((size_t)(i / ON_RLX_TILE) * RLX_PACKED_QUADS + w / 4) * ON_RLX_TILE + i % ON_RLX_TILE) * 4 + w % 4;
    };
    unsigned nthreads = std::thread::hardware_concurrency();
    if (nthreads == 0) nthreads = 1;
    if (nthreads > 16) nthreads = 16;
    if (ns < 65536) nthreads = 1;
    auto work = [&](int lo, int hi) {
        for (int i = lo; i < hi; i++) {
            RlxRegs r;
            rlx_seed_state(seeds[(size_t)i], nskip, r);
            unsigned d[RLX_SLOT_WORDS];
            rlx_pack(r, d);
            for (int k = 0; k < RLX_SLOT_WORDS; k++) st[at(k, i)] = d[k];
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nthreads; t++) {
        const int lo = (int)((long)ns * t / nthreads), hi = (int)((long)ns * (t + 1) / nthreads);
        if (nthreads == 1) work(lo, hi);
        else pool.emplace_back(work, lo, hi);
    }
    for (auto& t : pool) t.join();
    HIP_TRY(hipMemcpyAsync(c->rv.slot0, st.data(), st.size() * sizeof(unsigned), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->rv.slot1, 0, st.size() * sizeof(unsigned), c->stream));
    c->rv.cur = 0;
    c->rv.ctl = nullptr;
    HIP_TRY(hipMemsetAsync(c->counters, 0, 3 * sizeof(unsigned long long), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->rv.q = nskip / 24;
    c->rv.rem = nskip % 24;
    return QCD_OK;
}

template <typename F>
int launch_update(qcd_on_ctx* c, int parity)
{
    const dim3 grid(cdiv(c->g.vh, ON_BLOCK)), block(ON_BLOCK);
    if (c->p.lux_level == 3)
        hipLaunchKernelGGL((k_on_update<F, ON_BLOCK, 223>), grid, block, 0, c->stream, (F*)c->phi, c->g, parity,
                           params_of<F>(c), c->p.nhit, c->rv, c->wg_counts);
    else
        hipLaunchKernelGGL((k_on_update<F, ON_BLOCK, 0>), grid, block, 0, c->stream, (F*)c->phi, c->g, parity,
                           params_of<F>(c), c->p.nhit, c->rv, c->wg_counts);
    HIP_TRY(hipGetLastError());
    c->rv.cur ^= 1;   // every stream was used once and left its state in the other slot
    hipLaunchKernelGGL((k_on_count<1024>), dim3(1), dim3(1024), 0, c->stream, c->wg_counts, (int)grid.x, parity, c->counters);
    HIP_TRY(hipGetLastError());
    return QCD_OK;
}

int update(qcd_on_ctx* c, int parity)
{
    return c->p.precision ? launch_update<double>(c, parity) : launch_update<float>(c, parity);
}

template <typename F>
int launch_measure(qcd_on_ctx* c, const int step[4])
{
    const int4 st = make_int4(step[0], step[1], step[2], step[3]);
    hipLaunchKernelGGL((k_on_measure<F, ON_MEAS_BLOCK>), dim3(c->npartial), dim3(ON_MEAS_BLOCK), 0, c->stream,
                       (const F*)c->phi, c->g, params_of<F>(c), st, c->partial);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL((k_on_reduce<1024>), dim3(5), dim3(1024), 0, c->stream, c->partial, c->npartial, c->result_dev);
    HIP_TRY(hipGetLastError());
    return QCD_OK;
}

template <typename F>
int convert(qcd_on_ctx* c, void* table, bool to_engine)
{
    const dim3 grid(cdiv(2L * c->g.vh, ON_MEAS_BLOCK)), block(ON_MEAS_BLOCK);
    if (to_engine) hipLaunchKernelGGL((k_on_convert<F, true, ON_MEAS_BLOCK>), grid, block, 0, c->stream, (F*)c->phi, c->g, (F*)table);
    else hipLaunchKernelGGL((k_on_convert<F, false, ON_MEAS_BLOCK>), grid, block, 0, c->stream, (F*)c->phi, c->g, (F*)table);
    HIP_TRY(hipGetLastError());
    return QCD_OK;
}

int transfer(qcd_on_ctx* c, void* host, size_t bytes, bool up)
{
    const size_t want = (size_t)2 * c->g.vh * c->elem;
    if (!host) return fail(QCD_ERR_INVALID, "null host buffer");
    if (bytes != want) return fail(QCD_ERR_INVALID, "field is %zu bytes (%d sites of %zu), got %zu", want, 2 * c->g.vh, c->elem, bytes);
    void* dev = nullptr;
    HIP_TRY(hipMalloc(&dev, want));
    int rc = QCD_OK;
    hipError_t e = hipSuccess;
    if (up) {
        e = hipMemcpyAsync(dev, host, want, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) rc = c->p.precision ? convert<double>(c, dev, true) : convert<float>(c, dev, true);
    } else {
        rc = c->p.precision ? convert<double>(c, dev, false) : convert<float>(c, dev, false);
        if (rc == QCD_OK) e = hipMemcpyAsync(host, dev, want, hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(dev);
    if (e != hipSuccess) return fail(QCD_ERR_HIP, "field transfer: %s", hipGetErrorString(e));
    return rc;
}

}  // namespace

extern "C" {

int qcd_on_create(qcd_on_ctx** out, const qcd_on_params* p)
{
    if (!out || !p) return fail(QCD_ERR_INVALID, "null argument");
    *out = nullptr;
    for (int i = 0; i < 4; i++)
        if (p->L[i] < 2) return fail(QCD_ERR_INVALID, "lattice extent L[%d] = %d < 2", i, p->L[i]);
    if (p->L[1] & 1) return fail(QCD_ERR_INVALID, "Ny = %d must be even (checkerboard pairs sites 2h, 2h+1)", p->L[1]);
    // the field is stored by parity and a neighbour is looked up in the other parity's array: across
    // the periodic wrap of an odd extent the neighbour has the SAME parity
    if ((p->L[0] | p->L[2] | p->L[3]) & 1)
        return fail(QCD_ERR_UNSUPPORTED, "odd lattice extents are not supported (%d x %d x %d x %d): the checkerboard does not close", p->L[0],
                    p->L[1], p->L[2], p->L[3]);
    if (p->precision != 0 && p->precision != 1) return fail(QCD_ERR_UNSUPPORTED, "precision must be 0 (single) or 1 (double), got %d", p->precision);
    if (p->nhit < 1) return fail(QCD_ERR_INVALID, "NHIT = %d < 1", p->nhit);
    if (p->sk_group < 1) return fail(QCD_ERR_INVALID, "SKGROUP = %d < 1", p->sk_group);
    if (!(p->z > 0.0) || !(p->lambda > 0.0) || !(p->zeta > 0.0) || !(p->b > 0.0) || p->eta < 0.0)
        return fail(QCD_ERR_INVALID, "ON_Z, ON_LAMBDA, ON_ZETA, ON_B must be positive and ON_ETA non-negative");
    if (p->lux_level < 0 || p->lux_level > 4) return fail(QCD_ERR_INVALID, "RANLUX level %d not in 0..4", p->lux_level);
    if (p->rand_series == 0) return fail(QCD_ERR_INVALID, "RANDSERIES = 0 (time seed) must be resolved by the host");
    if (p->start_mode != QCD_START_HOT && p->start_mode != QCD_START_COLD)
        return fail(QCD_ERR_INVALID, "start mode %d is not hot (0) or cold (1)", p->start_mode);
    const long sites = (long)p->L[0] * p->L[1] * p->L[2] * p->L[3];
    if (sites > (1L << 30)) return fail(QCD_ERR_UNSUPPORTED, "lattice of %ld sites exceeds 2^30", sites);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (p->device < 0 || p->device >= ndev) return fail(QCD_ERR_INVALID, "device %d of %d", p->device, ndev);
    HIP_TRY(hipSetDevice(p->device));

    qcd_on_ctx* c = new qcd_on_ctx();
    memset(c, 0, sizeof(*c));
    c->p = *p;
    OnGeom& g = c->g;
    g.nx = p->L[0]; g.ny = p->L[1]; g.nz = p->L[2]; g.nt = p->L[3];
    g.nyh = g.ny / 2;
    g.slice_h = g.nyh * g.nz * g.nt;
    g.vh = g.nx * g.slice_h;
    {
        auto log2_of = [](int v) { int l = 0; while ((1 << l) < v) l++; return ((1 << l) == v) ? l : -1; };
        const int ly = log2_of(g.nyh), lz = log2_of(g.nz), lt = log2_of(g.nt);
        g.idx_shifts = (ly >= 0 && lz >= 0 && lt >= 0) ? ((lt << 16) | (lz << 8) | ly) : -1;
    }
    // model::model_lattice_init, O(N)/suncl/suncl.cpp:1244-1253
    c->sqrt_z_lambda_zeta = 0.25 * sqrt(p->z / (p->lambda * p->zeta));
    if (p->eta > 0.0) {
        double maxU1 = 1.0 - exp((1.0 - sqrt(1.0 + 4.0 * sqrt(2.0) * p->eta / sqrt(p->b))) / p->eta);
        const double maxU2 = 1.0 - exp((1.0 + sqrt(1.0 + 4.0 * sqrt(2.0) * p->eta / sqrt(p->b))) / p->eta);
        if (maxU1 < 0.0) maxU1 = maxU2;
        c->max_U = maxU1;
    } else {
        c->max_U = 1.0 - exp(-1.9999999999 / sqrt(p->b));
    }
    c->elem = p->precision ? sizeof(double) : sizeof(float);
    c->npartial = cdiv(2L * g.vh, ON_MEAS_BLOCK);

    int rc = QCD_OK;
    auto tryhip = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && rc == QCD_OK) rc = fail(e == hipErrorOutOfMemory ? QCD_ERR_NOMEM : QCD_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    };
    tryhip(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), "hipStreamCreate");
    tryhip(hipEventCreate(&c->ev0), "hipEventCreate");
    tryhip(hipEventCreate(&c->ev1), "hipEventCreate");
    tryhip(hipMalloc(&c->phi, (size_t)2 * g.vh * c->elem), "hipMalloc(field)");
    const size_t st_words = (size_t)RLX_SLOT_WORDS * cdiv(g.vh, ON_RLX_TILE) * ON_RLX_TILE;
    tryhip(hipMalloc(&c->rv.slot0, 2 * st_words * sizeof(unsigned)), "hipMalloc(ranlux)");
    c->rv.slot1 = c->rv.slot0 ? c->rv.slot0 + st_words : nullptr;
    c->rv.ns = g.vh;
    tryhip(hipMalloc(&c->counters, 3 * sizeof(unsigned long long)), "hipMalloc(counters)");
    tryhip(hipMalloc(&c->wg_counts, (size_t)2 * cdiv(g.vh, ON_BLOCK) * sizeof(unsigned long long)), "hipMalloc(workgroup counters)");
    tryhip(hipMalloc(&c->partial, (size_t)5 * c->npartial * sizeof(double)), "hipMalloc(partials)");
    tryhip(hipMalloc(&c->result_dev, 8 * sizeof(double)), "hipMalloc(result)");
    tryhip(hipHostMalloc(&c->result_host, 8 * sizeof(double)), "hipHostMalloc(result)");
    if (rc != QCD_OK) {
        qcd_on_destroy(c);
        return rc;
    }
    *out = c;
    return QCD_OK;
}

void qcd_on_destroy(qcd_on_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->p.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    (void)hipFree(c->phi);
    (void)hipFree(c->rv.slot0);
    (void)hipFree(c->counters);
    (void)hipFree(c->wg_counts);
    (void)hipFree(c->partial);
    (void)hipFree(c->result_dev);
    if (c->result_host) (void)hipHostFree(c->result_host);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int qcd_on_init_lattice(qcd_on_ctx* c)
{
    int rc = check(c, false);
    if (rc) return rc;
    rc = seed_streams(c);
    if (rc) return rc;
    const dim3 grid(cdiv(c->g.vh, ON_BLOCK)), block(ON_BLOCK);
    const int hot = c->p.start_mode == QCD_START_HOT;
    if (c->p.precision)
        hipLaunchKernelGGL((k_on_init<double, ON_BLOCK>), grid, block, 0, c->stream, (double*)c->phi, c->g, hot, (double)c->max_U, c->rv);
    else
        hipLaunchKernelGGL((k_on_init<float, ON_BLOCK>), grid, block, 0, c->stream, (float*)c->phi, c->g, hot, (float)c->max_U, c->rv);
    HIP_TRY(hipGetLastError());
    if (hot) {   // two quads per thread: one batch taken from every stream
        c->rv.cur ^= 1;
        const unsigned long long drawn = 8ull * (unsigned long long)c->g.vh;
        HIP_TRY(hipMemcpyAsync(c->counters, &drawn, sizeof(drawn), hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->initialized = true;
    return QCD_OK;
}

int qcd_on_update(qcd_on_ctx* c, int parity)
{
    int rc = check(c, true);
    if (rc) return rc;
    if (parity != 0 && parity != 1) return fail(QCD_ERR_INVALID, "parity %d is not 0 (even) or 1 (odd)", parity);
    return update(c, parity);
}

int qcd_on_sweep(qcd_on_ctx* c, int n)
{
    int rc = check(c, true);
    if (rc) return rc;
    if (n < 0) return fail(QCD_ERR_INVALID, "negative sweep count");
    for (int i = 0; i < n; i++) {
        if ((rc = update(c, 1)) != QCD_OK) return rc;
        if ((rc = update(c, 0)) != QCD_OK) return rc;
    }
    return QCD_OK;
}

int qcd_on_timed_sweeps(qcd_on_ctx* c, int n, double* elapsed_ms)
{
    int rc = check(c, true);
    if (rc) return rc;
    if (!elapsed_ms || n < 0) return fail(QCD_ERR_INVALID, "bad argument");
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    rc = qcd_on_sweep(c, n);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev1));
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, c->ev0, c->ev1));
    *elapsed_ms = t;
    return QCD_OK;
}

int qcd_on_measure(qcd_on_ctx* c, const int step[4], double out[5])
{
    int rc = check(c, true);
    if (rc) return rc;
    if (!step || !out) return fail(QCD_ERR_INVALID, "null argument");
    for (int d = 0; d < 4; d++)
        if (step[d] < 0 || step[d] > c->p.L[d]) return fail(QCD_ERR_INVALID, "correlator step[%d] = %d outside 0..%d", d, step[d], c->p.L[d]);
    rc = c->p.precision ? launch_measure<double>(c, step) : launch_measure<float>(c, step);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c->result_host, c->result_dev, 5 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const double sites = 2.0 * c->g.vh;
    for (int k = 0; k < 5; k++) out[k] = c->result_host[k] / sites;
    return QCD_OK;
}

int qcd_on_counters(qcd_on_ctx* c, unsigned long long out[3])
{
    int rc = check(c, false);
    if (rc) return rc;
    if (!out) return fail(QCD_ERR_INVALID, "null argument");
    HIP_TRY(hipMemcpyAsync(c->result_host, c->counters, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    memcpy(out, c->result_host, 3 * sizeof(unsigned long long));
    return QCD_OK;
}

int qcd_on_max_u(const qcd_on_ctx* c, double* out)
{
    if (!c || !out) return fail(QCD_ERR_INVALID, "null argument");
    *out = c->max_U;
    return QCD_OK;
}

int qcd_on_upload(qcd_on_ctx* c, const void* host_field, size_t bytes)
{
    int rc = check(c, false);
    if (rc) return rc;
    if (!c->initialized) {
        if ((rc = seed_streams(c)) != QCD_OK) return rc;
    }
    rc = transfer(c, const_cast<void*>(host_field), bytes, true);
    if (rc == QCD_OK) c->initialized = true;
    return rc;
}

int qcd_on_download(qcd_on_ctx* c, void* host_field, size_t bytes)
{
    int rc = check(c, true);
    if (rc) return rc;
    return transfer(c, host_field, bytes, false);
}

}  // extern "C"
