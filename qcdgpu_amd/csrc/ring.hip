// ring.hip -- one update pass over a ring of x-slabs, halo exchange included, below the C-ABI.
//
// The reference's host loop updates every part and then copies the edges through host memory,
// one part after the other (model::lattice_simulate with BIGLAT, SU(N)/suncl/suncl.cpp:5026-5138;
// model::UpdateEdges, :4005-4061).  Here a pass over (parity, direction) is issued for all parts
// of the ring in one call, without a host synchronisation:
//     1. every part updates the slice next to either face first (qcd_update_split: the rest of the
//        slab follows on the part's side lane, beside everything below);
//     2. every part packs what its neighbours' halos need: its last slice for the right
//        neighbour (all directions), its first slice for the left one (not for X links: nothing
//        reads U_x one slice to the right);
//     3. the messages travel -- device-to-device copies inside one process (QCD_RING_COPY), or
//        RCCL send/receive between ring neighbours on a stream of its own per part (QCD_RING_RCCL:
//        xGMI between GPUs; one ncclGroup per pass);
//     4. every part unpacks into its halo slices.
// The next pass of a part waits for its side lane by itself (every non-halo entry point does).
// RCCL is bound at run time (dlopen), so the library loads on machines without it.
#include "../../include/qcdgpu_amd.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace qcd {
int fail_msg(int code, const char* msg);
}

namespace {

int failf(int code, const char* fmt, const char* a = "", const char* b = "")
{
    char buf[512];
    snprintf(buf, sizeof(buf), fmt, a, b);
    return qcd::fail_msg(code, buf);
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

struct Part {
    qcd_ctx* ctx = nullptr;
    int device = 0, nx = 0;
    hipStream_t stream = nullptr;       // the context's stream at the time the ring was made
    char* out[2] = {nullptr, nullptr};  // message from the first (0) / last (1) owned slice
    char* in[2] = {nullptr, nullptr};   // message for the left (0) / right (1) halo
    hipEvent_t packed = nullptr, arrived = nullptr, consumed = nullptr;
    hipStream_t wire = nullptr;         // RCCL: the stream the part's sends and receives run on
    ncclComm_t comm = nullptr;
};

}  // namespace

struct qcd_ring {
    std::vector<Part> parts;
    int transport = QCD_RING_COPY;
    int rank0 = 0, world = 1;   // ring position of parts[0] and ring size (world > parts.size(): one part per process)
    size_t bytes = 0;
    unsigned long long passes = 0;
};

namespace {

int ring_alloc(qcd_ring* r)
{
    for (Part& p : r->parts) {
        RING_HIP(hipSetDevice(p.device));
        for (int s = 0; s < 2; s++) {
            RING_HIP(hipMalloc((void**)&p.out[s], r->bytes));
            RING_HIP(hipMalloc((void**)&p.in[s], r->bytes));
        }
        RING_HIP(hipEventCreateWithFlags(&p.packed, hipEventDisableTiming));
        RING_HIP(hipEventCreateWithFlags(&p.arrived, hipEventDisableTiming));
        RING_HIP(hipEventCreateWithFlags(&p.consumed, hipEventDisableTiming));
        if (r->transport == QCD_RING_RCCL) RING_HIP(hipStreamCreateWithFlags(&p.wire, hipStreamNonBlocking));
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
        const size_t b = qcd_halo_bytes(parts[k]);
        if (k && b != r->bytes) return failf(QCD_ERR_INVALID, "parts with different slice shapes");
        r->bytes = b;
        r->parts.push_back(p);
    }
    return QCD_OK;
}

// steps 1 and 2 of a pass for one part; `heatbath` false: overrelaxation (no split variant)
int part_update_and_pack(qcd_ring* r, Part& p, int parity, int dir, bool heatbath)
{
    int rc;
    void* st = nullptr;
    if ((rc = qcd_get_stream(p.ctx, &st))) return rc;   // the caller may have given the context another stream since
    p.stream = (hipStream_t)st;
    if (!heatbath) rc = qcd_overrelax(p.ctx, parity, dir);
    else if (p.nx >= 4) rc = qcd_update_split(p.ctx, parity, dir, p.nx - 1, 2);   // last and first slice, then the rest beside
    else rc = qcd_update(p.ctx, parity, dir);
    if (rc) return rc;
    RING_HIP(hipSetDevice(p.device));
    // the buffers are free: whoever read the last messages out of them has recorded `consumed`
    if (r->passes) RING_HIP(hipStreamWaitEvent(p.stream, p.consumed, 0));
    if ((rc = qcd_halo_pack(p.ctx, 1, dir, parity, p.out[1]))) return rc;
    if (dir != 0 && (rc = qcd_halo_pack(p.ctx, 0, dir, parity, p.out[0]))) return rc;
    RING_HIP(hipEventRecord(p.packed, p.stream));
    return QCD_OK;
}

int ring_pass(qcd_ring* r, int parity, int dir, bool heatbath)
{
    if (!r) return failf(QCD_ERR_INVALID, "null ring");
    if (parity != 0 && parity != 1) return failf(QCD_ERR_INVALID, "parity must be 0 or 1");
    if (dir < 0 || dir > 3) return failf(QCD_ERR_INVALID, "direction must be 0..3");
    const int n = (int)r->parts.size();
    int rc;
    for (Part& p : r->parts)
        if ((rc = part_update_and_pack(r, p, parity, dir, heatbath))) return rc;
    if (r->transport == QCD_RING_COPY) {
        // inside one process: the receiver's stream waits for the sender's packing and reads the
        // sender's buffer -- directly on the same device, through a peer copy otherwise
        for (int k = 0; k < n; k++) {
            Part& me = r->parts[k];
            Part& right = r->parts[(k + 1) % n];
            Part& left = r->parts[(k + n - 1) % n];
            RING_HIP(hipSetDevice(me.device));
            // left halo <- left neighbour's last slice
            RING_HIP(hipStreamWaitEvent(me.stream, left.packed, 0));
            const char* src0 = left.out[1];
            if (left.device != me.device) {
                RING_HIP(hipMemcpyPeerAsync(me.in[0], me.device, left.out[1], left.device, r->bytes, me.stream));
                src0 = me.in[0];
            }
            if ((rc = qcd_halo_unpack(me.ctx, 0, dir, parity, src0))) return rc;
            if (dir != 0) {
                // right halo <- right neighbour's first slice
                RING_HIP(hipStreamWaitEvent(me.stream, right.packed, 0));
                const char* src1 = right.out[0];
                if (right.device != me.device) {
                    RING_HIP(hipMemcpyPeerAsync(me.in[1], me.device, right.out[0], right.device, r->bytes, me.stream));
                    src1 = me.in[1];
                }
                if ((rc = qcd_halo_unpack(me.ctx, 1, dir, parity, src1))) return rc;
            }
            RING_HIP(hipEventRecord(me.arrived, me.stream));
        }
        // a part's buffers are free once both neighbours have read them
        for (int k = 0; k < n; k++) {
            Part& me = r->parts[k];
            RING_HIP(hipSetDevice(me.device));
            RING_HIP(hipStreamWaitEvent(me.stream, r->parts[(k + 1) % n].arrived, 0));
            RING_HIP(hipStreamWaitEvent(me.stream, r->parts[(k + n - 1) % n].arrived, 0));
            RING_HIP(hipEventRecord(me.consumed, me.stream));
        }
    } else {
        Rccl* nc = rccl();
        const size_t count = r->bytes / sizeof(double);
        for (Part& p : r->parts) {
            RING_HIP(hipSetDevice(p.device));
            RING_HIP(hipStreamWaitEvent(p.wire, p.packed, 0));
        }
        RING_NCCL(nc->GroupStart());
        for (int k = 0; k < n; k++) {
            Part& p = r->parts[k];
            const int me = r->rank0 + k;
            const int right = (me + 1) % r->world, left = (me + r->world - 1) % r->world;
            RING_NCCL(nc->Send(p.out[1], count, ncclDouble, right, p.comm, p.wire));
            RING_NCCL(nc->Recv(p.in[0], count, ncclDouble, left, p.comm, p.wire));
            if (dir != 0) {
                RING_NCCL(nc->Send(p.out[0], count, ncclDouble, left, p.comm, p.wire));
                RING_NCCL(nc->Recv(p.in[1], count, ncclDouble, right, p.comm, p.wire));
            }
        }
        RING_NCCL(nc->GroupEnd());
        for (Part& p : r->parts) {
            RING_HIP(hipSetDevice(p.device));
            RING_HIP(hipEventRecord(p.arrived, p.wire));
            RING_HIP(hipStreamWaitEvent(p.stream, p.arrived, 0));
            if ((rc = qcd_halo_unpack(p.ctx, 0, dir, parity, p.in[0]))) return rc;
            if (dir != 0 && (rc = qcd_halo_unpack(p.ctx, 1, dir, parity, p.in[1]))) return rc;
            // sends and receives of this pass are done (`arrived`), the incoming buffers are read
            RING_HIP(hipEventRecord(p.consumed, p.stream));
        }
    }
    r->passes++;
    return QCD_OK;
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
    for (Part& p : r->parts) {
        (void)hipSetDevice(p.device);
        if (p.stream) (void)hipStreamSynchronize(p.stream);
        if (p.wire) (void)hipStreamSynchronize(p.wire);
        if (p.comm && rccl()->lib) (void)rccl()->CommDestroy(p.comm);
        for (int s = 0; s < 2; s++) {
            (void)hipFree(p.out[s]);
            (void)hipFree(p.in[s]);
        }
        if (p.packed) (void)hipEventDestroy(p.packed);
        if (p.arrived) (void)hipEventDestroy(p.arrived);
        if (p.consumed) (void)hipEventDestroy(p.consumed);
        if (p.wire) (void)hipStreamDestroy(p.wire);
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

int qcd_ring_fill_halos(qcd_ring* r)
{
    if (!r) return failf(QCD_ERR_INVALID, "null ring");
    // every halo slice from its owner, without an update (needed once after a hot start)
    for (int dir = 0; dir < 4; dir++)
        for (int parity = 0; parity < 2; parity++) {
            const int n = (int)r->parts.size();
            if (r->transport == QCD_RING_COPY) {
                for (int k = 0; k < n; k++) {
                    int rc = qcd_halo_copy(r->parts[k].ctx, 1, r->parts[(k + 1) % n].ctx, 0, dir, parity);
                    if (rc) return rc;
                    if ((rc = qcd_halo_copy(r->parts[(k + 1) % n].ctx, 0, r->parts[k].ctx, 1, dir, parity))) return rc;
                }
            } else {
                return failf(QCD_ERR_UNSUPPORTED, "halo fill over RCCL: start cold or gid, or fill with the copy transport");
            }
        }
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
