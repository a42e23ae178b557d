// stub_capi.cpp -- TEST INFRASTRUCTURE ONLY.  A host-memory stand-in for the entry points of include/qcdgpu_amd.h that
// the C++ host (qcdgpu_amd/csrc/host/model.cpp) calls, so that the host's own code -- init-file tokenizer, run loop,
// result writer, state-file writer and reader -- runs on a machine without a GPU under AddressSanitizer and
// UndefinedBehaviorSanitizer.  No physics: a context keeps a link table in the reference device layout (cold start,
// upload and download copy it), "updates" count, measurements return fixed functions of that count.  Never linked
// into the product.
#include "../../include/qcdgpu_amd.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct qcd_ctx {
    qcd_params p;
    std::vector<double> table;   // rows x nloc x 4, reference layout
    std::vector<unsigned> rng;
    unsigned long long draws = 0;
    unsigned long long updates = 0;
    int share = 1;
};
struct qcd_ring {
    std::vector<qcd_ctx*> parts;
};

static thread_local std::string g_err;
static int fail(int code, const char* msg)
{
    g_err = msg;
    return code;
}
static size_t nloc(const qcd_ctx* c) { return (size_t)c->p.nx_local * c->p.L[1] * c->p.L[2] * c->p.L[3]; }
static int rows(const qcd_ctx* c) { return c->p.group == 3 ? 12 : 4; }
static size_t streams(const qcd_ctx* c) { return nloc(c) / 2 / (size_t)c->share; }

extern "C" {

const char* qcd_last_error(void) { return g_err.c_str(); }
int qcd_precision(void) { return 2; }
int qcd_rng_share_for(const int L[4], int min_nx_local)
{
    if ((L[0] | L[2] | L[3]) & 1) return 1;
    const long blocks = (long)min_nx_local * (L[1] / 2) * L[2] * L[3] / 64;
    if (L[3] % 4 == 0 && blocks / 4 >= 2048) return 4;
    return blocks / 2 >= 2048 ? 2 : 1;
}
int qcd_create(qcd_ctx** out, const qcd_params* p)
{
    if (!out || !p) return fail(QCD_ERR_INVALID, "null argument");
    for (int i = 0; i < 4; i++)
        if (p->L[i] < 2 || p->L[i] > 64) return fail(QCD_ERR_INVALID, "stub: lattice extent out of range");
    if (p->group != 2 && p->group != 3) return fail(QCD_ERR_INVALID, "stub: group");
    if (p->nx_local < 1 || p->x0 < 0 || p->x0 + p->nx_local > p->L[0]) return fail(QCD_ERR_INVALID, "stub: slab outside the lattice");
    if (p->precision != 2) return fail(QCD_ERR_UNSUPPORTED, "stub: double precision only");
    qcd_ctx* c = new qcd_ctx();
    c->p = *p;
    c->share = p->rng_share > 0 ? p->rng_share : 1;
    c->p.rng_share = c->share;
    *out = c;
    return QCD_OK;
}
void qcd_destroy(qcd_ctx* c) { delete c; }
int qcd_synchronize(qcd_ctx*) { return QCD_OK; }
int qcd_init_lattice(qcd_ctx* c)
{
    c->table.assign((size_t)rows(c) * nloc(c) * 4, 0.0);
    // cold start: u1 = v2 = 1 (SU(3): quad 0 .x of rows 0..3, quad 2 .y of rows 8..11; SU(2): .x)
    for (int dir = 0; dir < 4; dir++)
        for (size_t i = 0; i < nloc(c); i++) {
            c->table[((size_t)dir * nloc(c) + i) * 4 + 0] = 1.0;
            if (c->p.group == 3) c->table[((size_t)(8 + dir) * nloc(c) + i) * 4 + 1] = 1.0;
        }
    c->rng.assign((2 * 25 + 1) * streams(c), 7u);
    c->draws = c->updates = 0;
    return QCD_OK;
}
int qcd_update(qcd_ctx* c, int parity, int dir)
{
    if (parity < 0 || parity > 1 || dir < 0 || dir > 3) return fail(QCD_ERR_INVALID, "stub: pass");
    c->updates++;
    c->draws += 18 * nloc(c) / 2;
    return QCD_OK;
}
int qcd_overrelax(qcd_ctx* c, int, int) { c->updates++; return QCD_OK; }
int qcd_measure_plaquette_sums(qcd_ctx* c, double s[2])
{
    const double v = (double)nloc(c), g = c->p.group;
    s[0] = 3.0 * v * g / (1.0 + 0.01 * (double)c->updates);
    s[1] = 3.0 * v * g / (1.0 + 0.02 * (double)c->updates);
    return QCD_OK;
}
int qcd_measure_polyakov_sums(qcd_ctx* c, double s[4])
{
    const double n3 = (double)c->p.nx_local * c->p.L[1] * c->p.L[2];
    s[0] = n3; s[1] = 0.0; s[2] = n3; s[3] = n3;
    return QCD_OK;
}
int qcd_measure_fmunu_sums(qcd_ctx*, int, int, int, double s[12])
{
    for (int k = 0; k < 12; k++) s[k] = 0.001 * k;
    return QCD_OK;
}
int qcd_measure_plaquette_profile_sums(qcd_ctx* c, int axis, double* s)
{
    const int n = axis == 0 ? c->p.nx_local : c->p.L[axis];
    for (int i = 0; i < 2 * n; i++) s[i] = 1.0 + i;
    return QCD_OK;
}
int qcd_measure_polyakov_profile_sums(qcd_ctx* c, int axis, double* s) { return qcd_measure_plaquette_profile_sums(c, axis, s); }
int qcd_measure_wilson(qcd_ctx* c, int R, int T, double* out)
{
    if (R < 1 || T < 1) return fail(QCD_ERR_INVALID, "stub: loop size");
    *out = (double)c->p.group;
    return QCD_OK;
}
int qcd_upload(qcd_ctx* c, const void* t, size_t bytes, int rowsize)
{
    if (bytes != (size_t)rows(c) * nloc(c) * 32 || rowsize != (int)nloc(c)) return fail(QCD_ERR_INVALID, "stub: table size");
    c->table.resize(bytes / 8);
    memcpy(c->table.data(), t, bytes);
    return QCD_OK;
}
int qcd_download(qcd_ctx* c, void* t, size_t bytes, int rowsize)
{
    if (bytes != (size_t)rows(c) * nloc(c) * 32 || rowsize != (int)nloc(c)) return fail(QCD_ERR_INVALID, "stub: table size");
    if (c->table.size() * 8 != bytes) return fail(QCD_ERR_STATE, "stub: no lattice");
    memcpy(t, c->table.data(), bytes);
    return QCD_OK;
}
size_t qcd_rng_state_bytes(const qcd_ctx* c) { return 8 + (size_t)(2 * 25 + 1) * streams(c) * 4; }
int qcd_rng_state_download(qcd_ctx* c, void* h, size_t bytes)
{
    if (bytes != qcd_rng_state_bytes(c)) return fail(QCD_ERR_INVALID, "stub: state size");
    memcpy(h, &c->draws, 8);
    memcpy((char*)h + 8, c->rng.data(), bytes - 8);
    return QCD_OK;
}
int qcd_rng_state_upload(qcd_ctx* c, const void* h, size_t bytes)
{
    if (bytes != qcd_rng_state_bytes(c)) return fail(QCD_ERR_INVALID, "stub: state size");
    memcpy(&c->draws, h, 8);
    c->rng.resize((bytes - 8) / 4);
    memcpy(c->rng.data(), (const char*)h + 8, bytes - 8);
    return QCD_OK;
}
int qcd_halo_copy(qcd_ctx*, int, qcd_ctx*, int, int, int) { return QCD_OK; }
int qcd_ring_create(qcd_ring** out, qcd_ctx* const* parts, int n, int)
{
    qcd_ring* r = new qcd_ring();
    for (int k = 0; k < n; k++) r->parts.push_back(parts[k]);
    *out = r;
    return QCD_OK;
}
void qcd_ring_destroy(qcd_ring* r) { delete r; }
int qcd_ring_update(qcd_ring* r, int parity, int dir)
{
    for (qcd_ctx* c : r->parts) {
        const int rc = qcd_update(c, parity, dir);
        if (rc) return rc;
    }
    return QCD_OK;
}
int qcd_ring_sweep_chb2(qcd_ring* r, int n)
{
    for (int s = 0; s < n; s++)
        for (int parity = 1; parity >= 0; parity--)
            for (int dir = 0; dir < 4; dir++) qcd_ring_update(r, parity, dir);
    return QCD_OK;
}
int qcd_ring_overrelax(qcd_ring* r, int parity, int dir)
{
    for (qcd_ctx* c : r->parts) qcd_overrelax(c, parity, dir);
    return QCD_OK;
}
int qcd_device_name(int, char* buf, size_t n)
{
    snprintf(buf, n, "host-memory stub");
    return QCD_OK;
}
int qcd_set_profiling(qcd_ctx*, int) { return QCD_OK; }
int qcd_get_timing(qcd_ctx*, qcd_timing* t)
{
    memset(t, 0, sizeof(*t));
    return QCD_OK;
}

}  // extern "C"
