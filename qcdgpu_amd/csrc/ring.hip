// ring.hip -- one update pass over a ring of x-slabs, halo exchange included, below the C-ABI.
//
// The reference's host loop updates every part and then copies the edges through host memory,
// one part after the other (model::lattice_simulate with BIGLAT, SU(N)/suncl/suncl.cpp:5026-5138;
// model::UpdateEdges, :4005-4061).  Here a pass over (parity, direction) is issued for all parts
// of the ring in one call, without a host synchronisation:
//     1. on its context's stream every part updates its slab in ONE launch that starts with the
//        slices next to its two faces and writes a word when those are done and in memory
//        (qcd_update_signal_edges; slabs too small for that kernel, SU(2) and gid runs: two launches,
//        boundary slices first);
//     2. on a lane beside that stream the part waits for that word (hipStreamWaitValue32) -- from
//        then on its boundary slices can be read while the launch goes on with the interior;
//     3. the messages travel: last slice -> right neighbour's left halo (all directions), first
//        slice -> left neighbour's right halo (not for X links: nothing reads U_x one slice to the
//        right).  A boundary slice of the links of one direction and parity is six (SU(2): two)
//        contiguous runs, and so is the halo slice it goes to.  QCD_RING_COPY: the receiver's lane
//        copies them from lattice to lattice (one kernel for both halos on the same device, peer
//        copies otherwise); QCD_RING_RCCL: the runs are gathered into one buffer per neighbour, one
//        ncclSend / ncclRecv per neighbour (one ncclGroup per pass; xGMI between GPUs), and
//        scattered into the halo slices, all on the lane;
//     4. the context's stream waits for the halos behind the launch.
// Measured on a 32^4 slab that is its own neighbour (bare pass 122 us): 130 us per pass with the copy
// transport, 137 us over RCCL (sends to itself; RCCL's kernel takes 15 us alone and 60 us beside the
// update launch, whose memory traffic its handshakes queue behind).  Steps on the way (copy transport;
// over RCCL the first was 191 us): boundary slices and interior as launches on two streams, messages
// packed into buffers: 150 us -- the kernel of a large range keeps every CU to itself (a workgroup
// of eight wavefronts and 125 KB of LDS per CU), so the boundary launch ran before it, not beside
// it, and every hand-over between queues costs 10-20 us; two launches on one stream: 137 us (each
// launch has its ramp and its tail); one launch with a device-scope release fence per edge block:
// 149 us (each fence writes back an XCD's whole L2), with write-through stores instead: 128-130 us.
// RCCL is bound at run time (dlopen), so the library loads on machines without it.
#include "../../include/qcdgpu_amd.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

// the C++ names of the library live in namespace QCD_NS: qcd, or qcd_f32 in the single-precision build, so that both
// libraries can be loaded into one process and a profile tells their kernels apart
#ifndef QCD_NS
#define QCD_NS qcd
#endif
namespace QCD_NS {
int fail_msg(int code, const char* msg);
}

namespace {

int failf(int code, const char* fmt, const char* a = "", const char* b = "")
{
    char buf[512];
    snprintf(buf, sizeof(buf), fmt, a, b);
    return QCD_NS::fail_msg(code, buf);
}

#define RING_HIP(expr)                                                                  \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) return failf(QCD_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// the RCCL entry points the ring uses, resolved once per process
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl* rccl()
{
    static Rccl r;
    static bool tried = false;
    if (tried) return &r;
    tried = true;
    // a library of that name already in the process (PyTorch brings its own) is found first
    static const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names)
        if ((r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!r.lib) {
        r.error = std::string("librccl.so not found: ") + dlerror();
        return &r;
    }
    bool ok = true;
    auto sym = [&](const char* name) {
        void* p = dlsym(r.lib, name);
        if (!p) {
            ok = false;
            r.error = std::string("symbol missing in librccl: ") + name;
        }
        return p;
    };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    if (!ok) {
        dlclose(r.lib);
        r.lib = nullptr;
    }
    return &r;
}

#define RING_NCCL(expr)                                                                 \
    do {                                                                                \
        ncclResult_t e_ = (expr);                                                       \
        if (e_ != ncclSuccess) return failf(QCD_ERR_HIP, "%s failed: %s", #expr, rccl()->GetErrorString(e_)); \
    } while (0)

// a message where it lies in a lattice: `pieces` runs of `bytes`, `pitch` apart
struct Region {
    char* first = nullptr;
    size_t bytes = 0, pitch = 0;
    int pieces = 0;
};

struct Part {
    qcd_ctx* ctx = nullptr;
    Region region[2][2][4][2];          // [side][owned][direction][parity]: qcd_halo_region
    int device = 0, nx = 0;
    hipStream_t stream = nullptr;       // the context's stream: the update launches
    hipStream_t lane = nullptr;         // beside it: what moves the messages of a pass (copy transport: the copies themselves)
    // edge_done: the boundary slices are updated (stream, two-launch passes); ready: they can be read
    // by anyone on the device (lane);
    // halo_done: both halos are current (lane)
    hipEvent_t edge_done = nullptr, ready = nullptr, halo_done = nullptr;
    ncclComm_t comm = nullptr;
    char* out[2] = {nullptr, nullptr};  // RCCL: message from the first (0) / last (1) owned slice, packed
    char* in[2] = {nullptr, nullptr};   // RCCL: message for the left (0) / right (1) halo
    // one-launch passes: the update kernel writes the pass's number here when the boundary slices
    // are done and the lane waits for it (null: the device or this slab cannot, two launches instead)
    unsigned* signal = nullptr;
    unsigned signal_seq = 0;
};

int region_of(const Part& p, int side, int owned, int dir, int parity, Region& g)
{
    g = p.region[side][owned][dir][parity];   // asked for once, when the ring was made: a context's geometry does not change
    return QCD_OK;
}
int region_ask(const Part& p, int side, int owned, int dir, int parity, Region& g)
{
    void* f = nullptr;
    const int rc = qcd_halo_region(p.ctx, side, owned, dir, parity, &f, &g.bytes, &g.pitch, &g.pieces);
    g.first = (char*)f;
    return rc;
}

// Up to two messages moved by one launch (blockIdx.y picks the message): the runs of a slice to
// or from a contiguous buffer, or straight from a boundary slice into a halo slice on the same device.
struct CopyJob {
    const char* src;
    char* dst;
    size_t src_pitch, dst_pitch;   // in bytes
};
struct CopyJobs {
    CopyJob job[2];
};
// T: the unit a lane moves -- 16 bytes where runs and pitches allow it, else 8 (a stored complex number of the
// single-precision library is 8 bytes, so a run of an odd number of them is no whole number of 16-byte units)
template <class T>
__global__ __launch_bounds__(256) void k_slice_copy(const CopyJobs jobs, const int per_piece, const int pieces)
{
    const CopyJob j = jobs.job[blockIdx.y];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= per_piece * pieces) return;
    const int c = i / per_piece, r = i - c * per_piece;
    ((T*)(j.dst + (size_t)c * j.dst_pitch))[r] = ((const T*)(j.src + (size_t)c * j.src_pitch))[r];
}

int launch_slice_copy(const CopyJobs& jobs, int njobs, const Region& shape, hipStream_t stream)
{
    bool wide = shape.bytes % 16 == 0;
    for (int k = 0; k < njobs; k++)
        wide = wide && jobs.job[k].src_pitch % 16 == 0 && jobs.job[k].dst_pitch % 16 == 0 &&
               ((size_t)jobs.job[k].src | (size_t)jobs.job[k].dst) % 16 == 0;
    if (shape.bytes % 8 != 0) return failf(QCD_ERR_UNSUPPORTED, "the runs of a slice are not a multiple of 8 bytes long");
    const int unit = wide ? 16 : 8;
    const int per = (int)(shape.bytes / unit), n = per * shape.pieces;
    if (wide) hipLaunchKernelGGL(k_slice_copy<double2>, dim3((n + 255) / 256, njobs), dim3(256), 0, stream, jobs, per, shape.pieces);
    else hipLaunchKernelGGL(k_slice_copy<double>, dim3((n + 255) / 256, njobs), dim3(256), 0, stream, jobs, per, shape.pieces);
    RING_HIP(hipGetLastError());
    return QCD_OK;
}

}  // namespace

// One host thread per part, as the reference runs its parts under USE_OPENMP (one OpenMP thread per device,
// SU(N)/suncl/suncl.cpp:639-658, 2123-2128, 4018-4024): the handful of HIP calls a pass needs per part (26 us from one
// thread) are issued side by side, so the host's share of a pass does not grow with the number of parts.
// A step of a pass is a function of the part's number, run for all parts; the caller waits for all of them.
// Workers spin for the next step for a while (passes of a sweep follow one another at once), then sleep.
struct Pool {
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable go, done;
    std::atomic<unsigned long> step{0};     // number of the step posted last
    std::atomic<int> pending{0};            // parts that have not finished it
    std::function<int(int)> fn;
    std::vector<int> rc;
    std::vector<std::string> msg;
    bool stop = false;

    void worker(int k)
    {
        unsigned long seen = 0;
        for (;;) {
            unsigned long s = step.load(std::memory_order_acquire);
            for (int spin = 0; s == seen && spin < 20000; spin++) {
                __builtin_ia32_pause();
                s = step.load(std::memory_order_acquire);
            }
            if (s == seen) {
                std::unique_lock<std::mutex> lk(m);
                go.wait(lk, [&] { return stop || step.load(std::memory_order_acquire) != seen; });
                if (stop) return;
                s = step.load(std::memory_order_acquire);
            }
            {
                std::lock_guard<std::mutex> lk(m);   // `stop` and `fn` are written under the lock
                if (stop) return;
            }
            seen = s;
            rc[k] = fn(k);
            if (rc[k] != QCD_OK) msg[k] = qcd_last_error();   // the message is this thread's: hand it over
            if (pending.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                std::lock_guard<std::mutex> lk(m);
                done.notify_one();
            }
        }
    }
    void start(int n)
    {
        rc.assign(n, QCD_OK);
        msg.assign(n, std::string());
        for (int k = 0; k < n; k++) threads.emplace_back([this, k] { worker(k); });
    }
    // f(k) for every part k, on the parts' threads; the first failure's code (its message becomes the caller's)
    int run(const std::function<int(int)>& f)
    {
        const int n = (int)threads.size();
        {
            std::lock_guard<std::mutex> lk(m);
            fn = f;
            pending.store(n, std::memory_order_release);
            step.fetch_add(1, std::memory_order_acq_rel);
        }
        go.notify_all();
        for (int spin = 0; pending.load(std::memory_order_acquire) != 0 && spin < 200000; spin++) __builtin_ia32_pause();
        if (pending.load(std::memory_order_acquire) != 0) {
            std::unique_lock<std::mutex> lk(m);
            done.wait(lk, [&] { return pending.load(std::memory_order_acquire) == 0; });
        }
        for (int k = 0; k < n; k++)
            if (rc[k] != QCD_OK) return QCD_NS::fail_msg(rc[k], msg[k].c_str());
        return QCD_OK;
    }
    void shutdown()
    {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        go.notify_all();
        for (std::thread& t : threads) t.join();
        threads.clear();
    }
};

struct qcd_ring {
    std::vector<Part> parts;
    Pool* pool = nullptr;       // null: every step runs on the caller's thread, part after part
    int transport = QCD_RING_COPY;
    int rank0 = 0, world = 1;   // ring position of parts[0] and ring size (world > parts.size(): one part per process)
    size_t bytes = 0;
    unsigned long long passes = 0;
    std::atomic<unsigned long long> signalled{0};   // part-passes issued as one launch with the edge signal (counted from the parts' threads)
    int edge_override = 0;
    bool one_launch = true;
};

namespace {

int ring_alloc(qcd_ring* r)
{
    if (const char* ov = getenv("QCDGPU_AMD_RING_EDGE")) r->edge_override = atoi(ov);          // diagnostic overrides
    if (const char* ov = getenv("QCDGPU_AMD_RING_ONE_LAUNCH")) r->one_launch = atoi(ov) != 0;
    for (Part& p : r->parts) {
        RING_HIP(hipSetDevice(p.device));
        RING_HIP(hipEventCreateWithFlags(&p.edge_done, hipEventDisableTiming));
        RING_HIP(hipEventCreateWithFlags(&p.ready, hipEventDisableTiming));
        RING_HIP(hipEventCreateWithFlags(&p.halo_done, hipEventDisableTiming));
        // ONE stream of ordinary priority beside the context's: gathering, RCCL's sends and receives and
        // scattering follow one another anyway, and every hand-over between two queues costs 10-15 us.
        // A process has few hardware queues per device (four by default) and streams beyond that share
        // them: a stream for RCCL that shared the queue of the context's stream ran RCCL's kernel behind
        // the update launch instead of beside it.  (Streams of high priority have queues of their own
        // kind, but the wait for the edge word is a spinning kernel, and on a high-priority queue it kept
        // the update launch it waits for from starting: the pass hung.)
        RING_HIP(hipStreamCreateWithFlags(&p.lane, hipStreamNonBlocking));
        int can_wait = 0;
        if (r->one_launch && hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, p.device) == hipSuccess && can_wait) {
            if (hipExtMallocWithFlags((void**)&p.signal, 8, hipMallocSignalMemory) != hipSuccess) {
                (void)hipGetLastError();
                p.signal = nullptr;
            } else {
                RING_HIP(hipMemset(p.signal, 0, 8));
            }
        }
        if (r->transport == QCD_RING_RCCL) {
            // one send and one receive per neighbour and pass: RCCL moves few large messages much faster
            // than many small ones (a send per run of a slice, straight out of the lattice: 194 us per pass
            // of a 32^4 slab against 152 us), so the runs are gathered into a buffer first
            for (int s = 0; s < 2; s++) {
                RING_HIP(hipMalloc((void**)&p.out[s], r->bytes));
                RING_HIP(hipMalloc((void**)&p.in[s], r->bytes));
            }
        }
    }
    return QCD_OK;
}

int ring_add_parts(qcd_ring* r, qcd_ctx* const* parts, int nparts)
{
    for (int k = 0; k < nparts; k++) {
        qcd_params prm;
        void* st = nullptr;
        int rc = qcd_get_params(parts[k], &prm);
        if (rc) return rc;
        if ((rc = qcd_get_stream(parts[k], &st))) return rc;
        if (!prm.halo) return failf(QCD_ERR_INVALID, "a ring is made of contexts with halo slices (halo = 1)");
        Part p;
        p.ctx = parts[k];
        p.device = prm.device;
        p.nx = prm.nx_local;
        p.stream = (hipStream_t)st;
        for (int side = 0; side < 2; side++)
            for (int owned = 0; owned < 2; owned++)
                for (int dir = 0; dir < 4; dir++)
                    for (int parity = 0; parity < 2; parity++)
                        if ((rc = region_ask(p, side, owned, dir, parity, p.region[side][owned][dir][parity]))) return rc;
        const size_t b = qcd_halo_bytes(parts[k]);
        if (k && b != r->bytes) return failf(QCD_ERR_INVALID, "parts with different slice shapes");
        r->bytes = b;
        r->parts.push_back(p);
    }
    return QCD_OK;
}

// How many slices at each face go into the first launch of a pass that takes two.  The first
// launch holds the messages up, the second hides them: one slice per face is the minimum; for wide
// parts a few more make the first launch big enough for the kernel that full passes use.
int edge_slices(const qcd_ring* r, int nx)
{
    int a = nx >= 16 ? nx / 8 : 1;
    if (r->edge_override > 0) a = r->edge_override;
    if (2 * a > nx) a = nx / 2;
    return a;
}

// A pass for one part, as far as it needs nobody else.  On the context's stream: ONE launch that
// starts with the slices next to the two faces and says when they are done (qcd_update_signal_edges),
// or, where that is not to be had, two launches, boundary slices first.  On the lane: wait for
// that moment and record `ready`.  `heatbath` false: overrelaxation (one launch, then `ready`).
int part_update(qcd_ring* r, Part& p, int parity, int dir, bool heatbath)
{
    int rc;
    void* st = nullptr;
    if ((rc = qcd_get_stream(p.ctx, &st))) return rc;   // the caller may have given the context another stream since
    p.stream = (hipStream_t)st;
    RING_HIP(hipSetDevice(p.device));
    const int a = edge_slices(r, p.nx);
    bool signalled = false;
    if (heatbath && p.signal && p.nx >= 4) {
        rc = qcd_update_signal_edges(p.ctx, parity, dir, 1, p.signal, p.signal_seq + 1);
        if (rc == QCD_OK) {
            p.signal_seq++;
            r->signalled++;
            // passes of a part follow one another (the next launch waits for this pass's halos), so the
            // word holds this pass's number from the moment it is written until the lane has seen it
            RING_HIP(hipStreamWaitValue32(p.lane, p.signal, p.signal_seq, hipStreamWaitValueEq, 0xffffffffu));
            signalled = true;
        } else if (rc == QCD_ERR_UNSUPPORTED) {
            RING_HIP(hipFree(p.signal));   // this slab's passes do not run as the kernel that signals
            p.signal = nullptr;
        } else {
            return rc;
        }
    }
    if (!signalled) {
        if (heatbath && p.nx >= 4 && 2 * a < p.nx) {
            if ((rc = qcd_update_range(p.ctx, parity, dir, p.nx - a, 2 * a))) return rc;   // wraps: last a and first a slices
            RING_HIP(hipEventRecord(p.edge_done, p.stream));
            if ((rc = qcd_update_range(p.ctx, parity, dir, a, p.nx - 2 * a))) return rc;
        } else {
            if ((rc = heatbath ? qcd_update(p.ctx, parity, dir) : qcd_overrelax(p.ctx, parity, dir))) return rc;
            RING_HIP(hipEventRecord(p.edge_done, p.stream));
        }
        RING_HIP(hipStreamWaitEvent(p.lane, p.edge_done, 0));
    }
    RING_HIP(hipEventRecord(p.ready, p.lane));
    return QCD_OK;
}

// copy transport: the halos of `me` from the boundary slices of its neighbours, on me's lane
int pull(qcd_ring* r, Part& me, Part& left, Part& right, int dir, int parity, int njobs)
{
    // left halo <- left neighbour's last slice; right halo <- right neighbour's first slice (njobs = 1: the left
    // halo only)
    Part* from[2] = {&left, &right};
    Region src[2], dst[2];
    int rc;
    for (int s = 0; s < njobs; s++) {
        if ((rc = region_of(*from[s], 1 - s, 1, dir, parity, src[s]))) return rc;
        if ((rc = region_of(me, s, 0, dir, parity, dst[s]))) return rc;
        if (src[s].bytes != dst[s].bytes || src[s].pieces != dst[s].pieces) return failf(QCD_ERR_INVALID, "parts with different slice shapes");
        RING_HIP(hipStreamWaitEvent(me.lane, from[s]->ready, 0));
    }
    CopyJobs jobs = {};
    int same = 0;
    for (int s = 0; s < njobs; s++) {
        if (from[s]->device == me.device) {
            jobs.job[same++] = CopyJob{src[s].first, dst[s].first, src[s].pitch, dst[s].pitch};
        } else {
            for (int c = 0; c < src[s].pieces; c++)
                RING_HIP(hipMemcpyPeerAsync(dst[s].first + c * dst[s].pitch, me.device, src[s].first + c * src[s].pitch, from[s]->device, src[s].bytes,
                                            me.lane));
        }
    }
    if (same) return launch_slice_copy(jobs, same, src[0], me.lane);
    return QCD_OK;
}

// a step of a pass for every part: on the parts' own threads when the ring has them, else part after part
int for_parts(qcd_ring* r, const std::function<int(int)>& f)
{
    if (r->pool) return r->pool->run(f);
    for (int k = 0; k < (int)r->parts.size(); k++) {
        const int rc = f(k);
        if (rc) return rc;
    }
    return QCD_OK;
}

// The boundary slices of U_dir on the sites of one parity to the neighbours' halos, for every part; each part's
// `ready` event (lane) says from when on its boundary slices may be read and must be recorded before this is
// called.  sides = 2: both halos; 1: the left one only.
// The messages go from lattice to lattice: a boundary slice of the links (dir, parity) is a few
// contiguous runs (one per stored matrix element), and so is the halo slice it goes to.  The
// pass itself never reads U_dir on sites of this parity, so a halo may be overwritten while the
// part's own launch is still running; a boundary slice is not written again before the same
// (parity, dir) comes round, eight passes later, and no part gets more than two passes ahead
// of a neighbour.
int exchange(qcd_ring* r, int parity, int dir, int sides)
{
    const int n = (int)r->parts.size();
    int rc;
    // whatever comes next on a context's stream -- the next pass, a measurement -- finds the halos current
    auto finish = [r](Part& p) -> int {
        RING_HIP(hipEventRecord(p.halo_done, p.lane));
        RING_HIP(hipStreamWaitEvent(p.stream, p.halo_done, 0));
        return QCD_OK;
    };
    if (r->transport == QCD_RING_COPY)
        return for_parts(r, [=](int k) -> int {
            Part& me = r->parts[k];
            RING_HIP(hipSetDevice(me.device));
            const int rc2 = pull(r, me, r->parts[(k + n - 1) % n], r->parts[(k + 1) % n], dir, parity, sides);
            return rc2 ? rc2 : finish(me);
        });
    Rccl* nc = rccl();
    const size_t count = r->bytes / sizeof(double);
    // gather the runs of the last slice (to the right neighbour) and of the first (to the left)
    rc = for_parts(r, [=](int k) -> int {
        Part& p = r->parts[k];
        RING_HIP(hipSetDevice(p.device));
        Region reg[2];
        CopyJobs jobs = {};
        for (int s = 0; s < sides; s++) {
            const int rc2 = region_of(p, 1 - s, 1, dir, parity, reg[s]);
            if (rc2) return rc2;
            jobs.job[s] = CopyJob{reg[s].first, p.out[1 - s], reg[s].pitch, reg[s].bytes};
        }
        return launch_slice_copy(jobs, sides, reg[0], p.lane);
    });
    if (rc) return rc;
    // one group for all parts of this process, from the calling thread
    RING_NCCL(nc->GroupStart());
    for (int k = 0; k < n; k++) {
        Part& p = r->parts[k];
        const int me = r->rank0 + k;
        const int right = (me + 1) % r->world, left = (me + r->world - 1) % r->world;
        RING_NCCL(nc->Send(p.out[1], count, ncclDouble, right, p.comm, p.lane));
        RING_NCCL(nc->Recv(p.in[0], count, ncclDouble, left, p.comm, p.lane));
        if (sides == 2) {
            RING_NCCL(nc->Send(p.out[0], count, ncclDouble, left, p.comm, p.lane));
            RING_NCCL(nc->Recv(p.in[1], count, ncclDouble, right, p.comm, p.lane));
        }
    }
    RING_NCCL(nc->GroupEnd());
    return for_parts(r, [=](int k) -> int {
        Part& p = r->parts[k];
        RING_HIP(hipSetDevice(p.device));
        // (the lane gathers the next pass's messages behind this: the outgoing buffers are sent by then)
        Region reg[2];
        CopyJobs jobs = {};
        for (int s = 0; s < sides; s++) {
            const int rc2 = region_of(p, s, 0, dir, parity, reg[s]);
            if (rc2) return rc2;
            jobs.job[s] = CopyJob{p.in[s], reg[s].first, reg[s].bytes, reg[s].pitch};
        }
        const int rc2 = launch_slice_copy(jobs, sides, reg[0], p.lane);
        return rc2 ? rc2 : finish(p);
    });
}

int ring_pass(qcd_ring* r, int parity, int dir, bool heatbath)
{
    if (!r) return failf(QCD_ERR_INVALID, "null ring");
    if (parity != 0 && parity != 1) return failf(QCD_ERR_INVALID, "parity must be 0 or 1");
    if (dir < 0 || dir > 3) return failf(QCD_ERR_INVALID, "direction must be 0..3");
    // every part's `ready` event is recorded before any part waits for a neighbour's
    int rc = for_parts(r, [=](int k) -> int { return part_update(r, r->parts[k], parity, dir, heatbath); });
    if (rc) return rc;
    // after a pass over U_dir the right halo is stale for dir != X only: nothing reads U_x one slice to the right
    if ((rc = exchange(r, parity, dir, dir != 0 ? 2 : 1))) return rc;
    r->passes++;
    return QCD_OK;
}

// one host thread per part when the ring has more than one (QCDGPU_AMD_RING_THREADS = 0: never)
void ring_start_threads(qcd_ring* r)
{
    const char* ov = getenv("QCDGPU_AMD_RING_THREADS");
    if (r->parts.size() < 2 || (ov && atoi(ov) == 0)) return;
    r->pool = new Pool();
    r->pool->start((int)r->parts.size());
}

}  // namespace

extern "C" {

int qcd_ring_create(qcd_ring** out, qcd_ctx* const* parts, int nparts, int transport)
{
    if (!out || !parts || nparts < 1) return failf(QCD_ERR_INVALID, "bad arguments");
    *out = nullptr;
    if (transport != QCD_RING_COPY && transport != QCD_RING_RCCL) return failf(QCD_ERR_INVALID, "unknown transport");
    qcd_ring* r = new qcd_ring();
    r->transport = transport;
    r->world = nparts;
    int rc = ring_add_parts(r, parts, nparts);
    if (rc == QCD_OK && transport == QCD_RING_RCCL) {
        Rccl* nc = rccl();
        if (!nc->lib) rc = failf(QCD_ERR_UNSUPPORTED, "RCCL transport: %s", nc->error.c_str());
        std::vector<int> devs;
        for (Part& p : r->parts) devs.push_back(p.device);
        for (size_t i = 0; rc == QCD_OK && i < devs.size(); i++)
            for (size_t j = i + 1; j < devs.size(); j++)
                if (devs[i] == devs[j]) rc = failf(QCD_ERR_INVALID, "RCCL transport needs every part on a device of its own (use the copy transport)");
        if (rc == QCD_OK) {
            std::vector<ncclComm_t> comms(devs.size());
            ncclResult_t e = nc->CommInitAll(comms.data(), (int)devs.size(), devs.data());
            if (e != ncclSuccess) rc = failf(QCD_ERR_HIP, "ncclCommInitAll: %s", nc->GetErrorString(e));
            else
                for (size_t i = 0; i < devs.size(); i++) r->parts[i].comm = comms[i];
        }
    }
    if (rc == QCD_OK) rc = ring_alloc(r);
    if (rc != QCD_OK) {
        qcd_ring_destroy(r);
        return rc;
    }
    ring_start_threads(r);
    *out = r;
    return QCD_OK;
}

int qcd_rccl_unique_id(void* id, size_t bytes)
{
    if (!id || bytes != NCCL_UNIQUE_ID_BYTES) return failf(QCD_ERR_INVALID, "the identifier is 128 bytes");
    Rccl* nc = rccl();
    if (!nc->lib) return failf(QCD_ERR_UNSUPPORTED, "RCCL transport: %s", nc->error.c_str());
    ncclUniqueId u;
    RING_NCCL(nc->GetUniqueId(&u));
    memcpy(id, &u, NCCL_UNIQUE_ID_BYTES);
    return QCD_OK;
}

int qcd_ring_create_rank(qcd_ring** out, qcd_ctx* part, int rank, int world, const void* id, size_t bytes)
{
    if (!out || !part || !id || bytes != NCCL_UNIQUE_ID_BYTES || world < 1 || rank < 0 || rank >= world)
        return failf(QCD_ERR_INVALID, "bad arguments");
    *out = nullptr;
    Rccl* nc = rccl();
    if (!nc->lib) return failf(QCD_ERR_UNSUPPORTED, "RCCL transport: %s", nc->error.c_str());
    qcd_ring* r = new qcd_ring();
    r->transport = QCD_RING_RCCL;
    r->rank0 = rank;
    r->world = world;
    int rc = ring_add_parts(r, &part, 1);
    if (rc == QCD_OK) {
        hipError_t he = hipSetDevice(r->parts[0].device);
        ncclUniqueId u;
        memcpy(&u, id, NCCL_UNIQUE_ID_BYTES);
        if (he != hipSuccess) rc = failf(QCD_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(he));
        else {
            ncclResult_t e = nc->CommInitRank(&r->parts[0].comm, world, u, rank);
            if (e != ncclSuccess) rc = failf(QCD_ERR_HIP, "ncclCommInitRank: %s", nc->GetErrorString(e));
        }
    }
    if (rc == QCD_OK) rc = ring_alloc(r);
    if (rc != QCD_OK) {
        qcd_ring_destroy(r);
        return rc;
    }
    *out = r;
    return QCD_OK;
}

void qcd_ring_destroy(qcd_ring* r)
{
    if (!r) return;
    if (r->pool) {
        r->pool->shutdown();
        delete r->pool;
        r->pool = nullptr;
    }
    for (Part& p : r->parts) {
        (void)hipSetDevice(p.device);
        // the lane orders everything the ring issued; the context's stream belongs to the context, which may be gone
        // already if the caller destroyed it first (it should not: see the header)
        if (p.lane) (void)hipStreamSynchronize(p.lane);
        if (p.comm && rccl()->lib) (void)rccl()->CommDestroy(p.comm);
        if (p.edge_done) (void)hipEventDestroy(p.edge_done);
        if (p.ready) (void)hipEventDestroy(p.ready);
        for (int s = 0; s < 2; s++) {
            (void)hipFree(p.out[s]);
            (void)hipFree(p.in[s]);
        }
        if (p.halo_done) (void)hipEventDestroy(p.halo_done);
        if (p.lane) (void)hipStreamDestroy(p.lane);
        if (p.signal) (void)hipFree(p.signal);
    }
    delete r;
}

int qcd_ring_update(qcd_ring* r, int parity, int dir) { return ring_pass(r, parity, dir, true); }

int qcd_ring_overrelax(qcd_ring* r, int parity, int dir) { return ring_pass(r, parity, dir, false); }

int qcd_ring_sweep(qcd_ring* r, int n)
{
    if (n < 0) return failf(QCD_ERR_INVALID, "negative sweep count");
    for (int s = 0; s < n; s++)
        for (int parity = 1; parity >= 0; parity--)   // SU(N)/suncpp/suncpp.h:133-141: odd X,Y,Z,T then even
            for (int dir = 0; dir < 4; dir++) {
                const int rc = ring_pass(r, parity, dir, true);
                if (rc) return rc;
            }
    return QCD_OK;
}

// The reference's CHB2 build (SU(N)/suncl/suncl.cpp:4683-4729): in its working cycles every part runs the eight passes of
// a sweep on its own and the edges of all four directions are copied once, behind the sweep.  Inside the sweep a part
// reads halos that are up to a sweep old, so the chain differs from the uncut lattice's at the slab faces (it is the
// reference's variant, reproduced as it is); in exchange a sweep costs one round of messages instead of eight.
int qcd_ring_sweep_chb2(qcd_ring* r, int n)
{
    if (!r) return failf(QCD_ERR_INVALID, "null ring");
    if (n < 0) return failf(QCD_ERR_INVALID, "negative sweep count");
    for (int s = 0; s < n; s++) {
        int rc = for_parts(r, [=](int k) -> int {
            Part& p = r->parts[k];
            void* st = nullptr;
            int rc2 = qcd_get_stream(p.ctx, &st);
            if (rc2) return rc2;
            p.stream = (hipStream_t)st;
            RING_HIP(hipSetDevice(p.device));
            for (int parity = 1; parity >= 0; parity--)      // SU(N)/suncpp/suncpp.h:133-141: odd X,Y,Z,T then even
                for (int dir = 0; dir < 4; dir++)
                    if ((rc2 = qcd_update(p.ctx, parity, dir))) return rc2;
            RING_HIP(hipEventRecord(p.edge_done, p.stream));
            RING_HIP(hipStreamWaitEvent(p.lane, p.edge_done, 0));
            RING_HIP(hipEventRecord(p.ready, p.lane));
            return QCD_OK;
        });
        if (rc) return rc;
        for (int dir = 0; dir < 4; dir++)
            for (int parity = 0; parity < 2; parity++)
                if ((rc = exchange(r, parity, dir, 2))) return rc;
        r->passes += 8;
    }
    return QCD_OK;
}

int qcd_ring_fill_halos(qcd_ring* r)
{
    if (!r) return failf(QCD_ERR_INVALID, "null ring");
    // Every halo slice from its owner, without an update: needed once after a hot start or an upload (the reference:
    // UpdateEdges for all four directions after the start kernels, SU(N)/suncl/suncl.cpp:4835-4903).  The same
    // exchange as behind a pass, over either transport, for all directions, both parities and both halos.
    int rc = for_parts(r, [=](int k) -> int {
        Part& p = r->parts[k];
        void* st = nullptr;
        const int rc2 = qcd_get_stream(p.ctx, &st);
        if (rc2) return rc2;
        p.stream = (hipStream_t)st;
        RING_HIP(hipSetDevice(p.device));
        RING_HIP(hipEventRecord(p.edge_done, p.stream));    // whatever wrote the links is in this stream
        RING_HIP(hipStreamWaitEvent(p.lane, p.edge_done, 0));
        RING_HIP(hipEventRecord(p.ready, p.lane));
        return QCD_OK;
    });
    if (rc) return rc;
    for (int dir = 0; dir < 4; dir++)
        for (int parity = 0; parity < 2; parity++)
            if ((rc = exchange(r, parity, dir, 2))) return rc;
    return QCD_OK;
}

int qcd_ring_one_launch_passes(qcd_ring* r, unsigned long long* n)
{
    if (!r || !n) return failf(QCD_ERR_INVALID, "null argument");
    *n = r->signalled.load();
    return QCD_OK;
}

int qcd_ring_synchronize(qcd_ring* r)
{
    if (!r) return failf(QCD_ERR_INVALID, "null ring");
    for (Part& p : r->parts) {
        const int rc = qcd_synchronize(p.ctx);
        if (rc) return rc;
    }
    return QCD_OK;
}

}  // extern "C"
