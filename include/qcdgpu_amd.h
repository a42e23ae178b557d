/*
 * qcdgpu_amd.h -- C-ABI of the MI355X SU(N) sweep engine (libqcdgpu_amd.so).
 *
 * The reference has no FFI: its de-facto boundary is the pair of C++ classes
 * model_CL::model (run API, SU(N)/suncl/suncl.h:385-399) and GPU_CL::GPU (device service,
 * SU(N)/clinterface/clinterface.h:219-251).  Each entry point below replaces the kernel_run /
 * buffer_map call sites of model::lattice_simulate that are cited beside it.  Plain C types
 * only; every function returns a status (0 = QCD_OK) instead of exiting the process the way
 * the reference's OpenCL error handler does (SU(N)/clinterface/clinterface.cpp:225-284).
 *
 * Threading: one context is driven by one host thread at a time.  All work of a context is
 * enqueued on one HIP stream (its own, or the one given to qcd_set_stream); functions that
 * return values to the host synchronise that stream, the others are asynchronous.
 */
#ifndef QCDGPU_AMD_H
#define QCDGPU_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qcd_ctx qcd_ctx;

enum {
    QCD_OK = 0,
    QCD_ERR_INVALID = 1,      /* bad argument / unsupported configuration */
    QCD_ERR_HIP = 2,          /* a HIP runtime call failed (see qcd_last_error) */
    QCD_ERR_NOMEM = 3,
    QCD_ERR_UNSUPPORTED = 4,
    QCD_ERR_STATE = 5         /* call out of order (e.g. update before init) */
};

/* start modes: model::ints, SU(N)/suncl/suncl.h:64-68 (INTS key of init.dat) */
enum { QCD_START_HOT = 0, QCD_START_COLD = 1, QCD_START_GID = 2 };

/* run parameters <-> public fields of model (SU(N)/suncl/suncl.h:88-167) and of PRNG
 * (SU(N)/random/random.h).  L is the FULL lattice; a context owns the x-slab
 * [x0, x0 + nx_local) of it (the reference's SubLattice, suncl.h:470-699).  A single-GPU run
 * has x0 = 0, nx_local = L[0], halo = 0. */
typedef struct {
    int      group;        /* lattice_group: 2 or 3 */
    int      nd;           /* lattice_nd: 4 */
    int      L[4];         /* lattice_full_size: Nx, Ny, Nz, Nt.  Ny even (coord_work.cpp:151-175); odd Nx, Nz, Nt for a
                              context that owns the whole lattice (halo = 0) only */
    int      precision;    /* 2 = double, 1 = single, 3 = mixed (model::convert_precision_to_uint,
                              SU(N)/suncl/suncl.cpp:732-737); must be what qcd_precision() reports (3 counts as 1) */
    double   beta;         /* BETA */
    int      nhit;         /* NHIT: max heat-bath trials per SU(2) subgroup */
    int      start_mode;   /* QCD_START_* (INTS) */
    int      lux_level;    /* RANLUX luxury level 0..4 (PRNG = RANLUX0..RANLUX4) */
    unsigned rand_series;  /* RANDSERIES (must be non-zero: the time-based seed is the host's job) */
    int      device;       /* HIP device ordinal (DEVICE) */
    int      x0;           /* first global x slice owned by this context */
    int      nx_local;     /* number of x slices owned (SubLattice::Nx); even, or the whole lattice */
    int      halo;         /* 0: periodic in x inside the context; 1: two halo slices, filled by
                              qcd_halo_unpack (BigLattice geometry, suncl.cpp:2131-2148) */
    double   phi, omega;   /* PHI, OMEGA: twisted boundary condition (-D TBC when one of them is not 0,
                              suncl.cpp:3286).  Every READ of U_y at x = Nx-1 by an update or a measurement returns
                              Omega U_y, Omega = diag(e^{i(phi+omega)/2}, e^{i(omega-phi)/2}, e^{-i omega}) for SU(3)
                              (su3_matrix_memory.cl:57-98, su3cpu.cpp:165-197), diag(e^{i phi/2}, e^{-i phi/2}) for SU(2)
                              (su2_matrix_memory.cl:63-88); an updated link is stored as it comes out of the heat-bath
                              (suncl.cl:525-536), re-unitarisation and the link tables see the stored values.  halo = 0 only. */
    int      rng_share;    /* sites of one parity per RANLUX stream: 1, 2 or 4; 0 = chosen from the FULL lattice (4 or 2 when
                              it is large enough to keep the device busy with that many fewer independent streams, else 1).
                              The sites (x, y-pair, z, t + k Nt/rng_share), k = 0 .. rng_share-1, are updated by one lane one
                              after the other from one stream, whose state then travels once per pass instead of rng_share
                              times.  The reference's device generator likewise serves many sites per instance
                              (prns[thread + trial * PRNGSTEP], SU(N)/suncl/su3_update_cl.cl:597-649 with PRNG_instances
                              streams, SU(N)/random/random.cpp:216-223).  The map is a function of global coordinates: every
                              part of a cut lattice must be given the same value (0 in all of them does).  Nt % rng_share = 0. */
} qcd_params;

typedef struct {
    unsigned long long update_launches;   /* update_{even,odd}_{X,Y,Z,T} launches */
    double             update_ms;         /* device time of those launches (profiling on) */
    unsigned long long measure_launches;
    double             measure_ms;
    unsigned long long halo_launches;
    double             halo_ms;
} qcd_timing;

/* model::model + lattice_init + lattice_create_buffers (suncl.cpp:142-214, 672-706, 3877-3991):
 * selects the device, allocates the link table, the RANLUX streams and the reduction buffers. */
int  qcd_create(qcd_ctx **out, const qcd_params *params);
void qcd_destroy(qcd_ctx *ctx);

/* message of the last failing call on this thread */
const char *qcd_last_error(void);

/* The precision this library was built for: 2 = double (libqcdgpu_amd.so), 1 = single
 * (libqcdgpu_amd_f32.so: links, staples and the heat-bath in float, lattice sums in double).  The reference
 * fixes the precision when it builds its kernels (-D PRECISION, SU(N)/suncl/suncl.cpp:3274-3301,
 * SU(N)/kernel/complex.h:57-92) and sets it in its source, SU(N)/QCDGPU.cpp:178-179; here it is the
 * choice of the library a program loads.  Both libraries export the same entry points; link tables
 * (qcd_upload/qcd_download, the halo messages) are in the library's precision. */
int qcd_precision(void);

/* use a caller-owned hipStream_t (passed as void*) for all subsequent work; NULL is the device's
 * default stream.  qcd_reset_stream goes back to the context's own (non-blocking) stream. */
int qcd_set_stream(qcd_ctx *ctx, void *hip_stream);
int qcd_reset_stream(qcd_ctx *ctx);
int qcd_synchronize(qcd_ctx *ctx);
/* the stream the context's work is enqueued on at present (a hipStream_t), and the parameters it was created with */
int qcd_get_stream(qcd_ctx *ctx, void **hip_stream);
int qcd_get_params(const qcd_ctx *ctx, qcd_params *out);

/* kernel_run(sun_init_id) (suncl.cpp:5243-5252; kernels lattice_init_cold / lattice_init_gid,
 * SU(N)/suncl/suncl.cl:163-347): cold or gid start; also resets the RANLUX streams */
int qcd_init_lattice(qcd_ctx *ctx);

/* one kernel_run(sun_update_<parity>_<dir>_id) together with the PRNG0->produce() in front of
 * it (suncl.cpp:5359-5374): heat-bath update of all links of direction dir (0..3 = X,Y,Z,T) on
 * the sites of one parity (0 even, 1 odd), re-unitarisation included */
int qcd_update(qcd_ctx *ctx, int parity, int dir);

/* The value rng_share = 0 stands for: sites per RANLUX stream for a lattice L cut into parts of at least
 * min_nx_local x-slices (min_nx_local = L[0]: uncut, what qcd_create itself assumes).  A host that cuts the lattice
 * passes the result to every part, so that thin parts are not left with too few streams to fill a device. */
int qcd_rng_share_for(const int L[4], int min_nx_local);

/* Which of the library's two heat-bath kernels a pass over xs_count owned slices runs as: *pair_kernel = 1 for the
 * gather / heat-bath wavefront-pair kernel (SU(3), per-thread streams, at least two rounds of its teams), 0 for the
 * kernel with one wavefront per block of sites.  No counterpart in the reference (one kernel per pass there,
 * suncl.cpp:5359-5374); for programs that label measurements by kernel. */
int qcd_update_kernel(qcd_ctx *ctx, int xs_count, int *pair_kernel);

/* n full sweeps in the reference order: odd X,Y,Z,T then even X,Y,Z,T (suncpp.h:133-141).
 * Only valid for halo = 0 (a slab needs the halo exchange between passes). */
int qcd_sweep(qcd_ctx *ctx, int n);

/* qcd_update restricted to xs_count owned x-slices starting at xs_begin (wrapping around the owned
 * range).  No counterpart in the reference, which updates a part as a whole and then copies its
 * edges (suncl.cpp:5026-5138): a slab updates its boundary slices first and sends them while the
 * interior is being updated.  The ranges of one pass must tile the slab exactly once: a range whose
 * slices have been through different numbers of passes is refused (QCD_ERR_STATE), because all random
 * streams of a launch are read from the same one of their two state slots. */
int qcd_update_range(qcd_ctx *ctx, int parity, int dir, int xs_begin, int xs_count);

/* One pass in two concurrent launches: the owned slices [first_begin, first_begin + first_count)
 * (wrapping) on the context's stream, all the others on a stream of the context's own beside them.
 * Calls that follow on the context's stream are ordered after the FIRST part only as long as they
 * are halo calls (qcd_halo_pack / _unpack / _copy); every other entry point waits for the rest as
 * well.  So a slab updates its boundary slices, packs and sends them while the interior is still
 * being updated, without the two launches having to run one after the other.  No counterpart in
 * the reference (see qcd_update_range). */
int qcd_update_split(qcd_ctx *ctx, int parity, int dir, int first_begin, int first_count);

/* Overrelaxation (microcanonical reflection in the three SU(2) subgroups, same pass structure
 * and halo rule as qcd_update).  The reference has no overrelaxation: no call site to replace,
 * and no reference output to pin it to -- it is checked by the invariance of the action. */
int qcd_overrelax(qcd_ctx *ctx, int parity, int dir);
int qcd_overrelax_sweep(qcd_ctx *ctx, int n);

/* kernel_run(sun_GramSchmidt_id) (suncl.cpp:5376; kernel lattice_GramSchmidt, suncl.cl:349-396) */
int qcd_reunitarize(qcd_ctx *ctx);

/* measurement + reduce kernel pairs (suncl.cpp:5425-5483):
 *   plaquette: out = Plq_spat, Plq_temp, Plq_total (not divided by N; cold start gives N)
 *   action:    out = S_spat, S_temp, S_total = beta * (1 - Plq / N)
 *   polyakov:  out = <Re P>/N, <Im P>/N, <|P|^2>/N^2, <|P|^4>/N^4
 * The *_sums variants return this context's raw sums (for slabs: add over contexts, then
 * normalise with the FULL volume): plaquette sums = (sum_spat, sum_temp) of ReTr over the 3+3
 * planes; polyakov sums = (sum Re P, sum Im P, sum |P|^2, sum |P|^4). */
int qcd_measure_plaquette(qcd_ctx *ctx, double out[3]);
int qcd_measure_action(qcd_ctx *ctx, double out[3]);
int qcd_measure_polyakov(qcd_ctx *ctx, double out[4]);
int qcd_measure_plaquette_sums(qcd_ctx *ctx, double sums[2]);
int qcd_measure_polyakov_sums(qcd_ctx *ctx, double sums[4]);

/* No counterpart in the reference.  For callers that cannot allocate: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  The plaquette sums of
 * qcd_measure_plaquette_sums computed by the same kernels, with the per-workgroup partial sums in a
 * static array of the library's code object (1 MiB per device, held until the process ends)
 * instead of in allocated memory.  The call allocates nothing, does not synchronise and copies
 * nothing to the host: sums_dev (device-accessible memory of the caller, two doubles) receives
 * (sum_spat, sum_temp) once the context's stream has run the two kernels.  Limit: at most
 * 65536 * 256 = 2^24 owned sites per context (QCD_ERR_UNSUPPORTED beyond).  All calls on one
 * device share that device's array: calls that overlap in time on one device race. */
int qcd_measure_plaquette_sums_static(qcd_ctx *ctx, double *sums_dev);

/* kernel_run(sun_measurement_plq_id) built with FMUNU or F0MU + its reduce kernel (suncl.cpp:5478-5483;
 * kernel lattice_measurement_plq, sun_measurements_cl.cl:712-1175; CPU twin su3cpu.cpp:1212-1381):
 * F_re = Im Tr(P lambda_a)/2, F_im = -Re Tr(P lambda_a)/2 of the plaquette matrix P, per plane.
 * field 0 = planes xy, xz, yz (FMUNU, "H field"), 1 = xt, yt, zt (F0MU, "E field"); index1 in
 * {1,2,3,4}, index2 in {5,6,7,8} in the reference (FMUNU1.. keys), any 1..8 here; SU(2): Pauli
 * matrices 1..3 (the reference: index1 3 or 1, index2 2; su2cpu.cpp:716-870).  out = for
 * (index1, index2) x three planes x (re, im).  qcd_measure_fmunu: averages over this context's
 * sites; _sums: raw sums (slabs: add over contexts, divide by the full volume). */
int qcd_measure_fmunu(qcd_ctx *ctx, int field, int index1, int index2, double out[12]);
int qcd_measure_fmunu_sums(qcd_ctx *ctx, int field, int index1, int index2, double sums[12]);

/* kernel_run(sun_action_diff_{x,y,z}_id) / kernel_run(sun_polyakov_diff_{x,y,z}_id) with their reduce
 * kernels (suncl.cpp:5432-5449, 5458-5476; kernels lattice_action_diff_*, sun_measurements_cl.cl:186-616,
 * lattice_polyakov_diff_*, polyakov.cl:142-402): profiles along a space axis (0 x, 1 y, 2 z).
 * sums[2*i], sums[2*i+1] = raw (spat, temp) ReTr plaquette sums resp. (Re P, Im P) sums over the
 * owned sites with coordinate i along the axis; i runs over nx_local (axis 0), Ny or Nz values.
 * The host turns them into S_s/S_t resp. PL/PL_im per plane (suncl.cpp:1116-1260).
 * Cost: a profile is summed from per-site values that qcd_measure_plaquette(_sums) resp.
 * qcd_measure_polyakov(_sums) leave on the device; they stay valid while only measurement calls are
 * made on the context, otherwise the profile call recomputes them.  The first plaquette profile of a
 * context allocates the per-site array (16 bytes per owned site). */
int qcd_measure_plaquette_profile_sums(qcd_ctx *ctx, int axis, double *sums);
int qcd_measure_polyakov_profile_sums(qcd_ctx *ctx, int axis, double *sums);

/* kernel_run(sun_measurement_wilson_id) + sun_wilson_loop_reduce_id (suncl.cpp:5266-5275): the
 * R x T Wilson loop averaged over sites and the three space-time planes (denominator 3V,
 * suncl.cpp:1606-1611).  One context owning the whole lattice only. */
int qcd_measure_wilson(qcd_ctx *ctx, int R, int T, double *out);

/* buffer_map / COPY_HOST_PTR of lattice_table (suncl.cpp:5498-5504, 3877-3991): the link table
 * of this context's slab in the REFERENCE device layout and the library's precision (qcd_precision): rows of
 * double4, or of float4 in the single-precision build (hgpu_float4, SU(N)/suncl/model.cl:140-162),
 * element [gindex + row*rowsize], gindex = y + z*Ny + t*Ny*Nz + x_local*Ny*Nz*Nt;
 * SU(3): 12 rows, row = quad*4 + dir, quads (Re u1,Re u2,Re u3,Re v3), (Im u1,Im u2,Im u3,Im v3),
 * (Re v1,Re v2,Im v1,Im v2) (SU(N)/suncl/su3_matrix_memory.cl:42-176); SU(2): 4 rows, row = dir,
 * (Re u1,Re u2,Im u1,Im u2) (su2_matrix_memory.cl:55-142).  bytes = rows*rowsize*32 (double), rows*rowsize*16 (single). */
int qcd_upload(qcd_ctx *ctx, const void *host_table, size_t bytes, int rowsize);
int qcd_download(qcd_ctx *ctx, void *host_table, size_t bytes, int rowsize);

/* generator part of lattice_save_state / lattice_load_state (suncl.cpp:1857-2079).  The reference
 * restores its generator by replaying PRNG0->produce() PRNG_counter times (suncl.cpp:5349-5351);
 * per-thread streams are saved instead: [draws u64][ctl ns][slot0 25*ns][slot1 25*ns], 32-bit
 * words, ns = owned half-sites, the slots in word-major order [w][stream] (24 words of 24 bits and
 * the carry; the device keeps them packed, these calls convert).  Upload after qcd_upload or
 * qcd_init_lattice. */
size_t qcd_rng_state_bytes(const qcd_ctx *ctx);
int qcd_rng_state_download(qcd_ctx *ctx, void *host, size_t bytes);
int qcd_rng_state_upload(qcd_ctx *ctx, const void *host, size_t bytes);

/* model::UpdateEdges(dir) (suncl.cpp:4005-4061), split into its two halves so that the transport
 * between contexts (peer copy, RCCL send/recv) is the caller's.  A message carries the links of
 * ONE direction on the sites of ONE parity of one boundary x-slice:
 *   qcd_halo_pack  (side 0): first owned slice  -> buffer (goes to the LEFT neighbour's side 1)
 *   qcd_halo_pack  (side 1): last owned slice   -> buffer (goes to the RIGHT neighbour's side 0)
 *   qcd_halo_unpack(side 0): buffer -> left halo slice;  (side 1): buffer -> right halo slice
 * dev_buf is device memory of qcd_halo_bytes() bytes. */
size_t qcd_halo_bytes(const qcd_ctx *ctx);
int    qcd_halo_pack(qcd_ctx *ctx, int side, int dir, int parity, void *dev_buf);
int    qcd_halo_unpack(qcd_ctx *ctx, int side, int dir, int parity, const void *dev_buf);
/* One pass over the links (parity, dir) of a slab in ONE launch that starts with the `edge` slices
 * next to each face and goes on with the interior; as soon as the edge slices are stored and
 * visible to the device it writes `value` to *signal, a 32-bit word of signal memory
 * (hipExtMallocWithFlags(.., 8, hipMallocSignalMemory)) on which another stream waits with
 * hipStreamWaitValue32 before it packs the boundary slices -- the messages of a pass then travel
 * beside the rest of the same launch.  QCD_ERR_UNSUPPORTED (and nothing launched) when this pass
 * would not run as the kernel that can signal (SU(2), gid mode, small or ragged slabs): the caller
 * falls back to two launches (qcd_update_range).  No counterpart in the reference, whose parts are
 * updated whole before their edges are copied (suncl.cpp:5026-5138). */
int    qcd_update_signal_edges(qcd_ctx *ctx, int parity, int dir, int edge, void *signal, unsigned value);

/* Where a message lies in the lattice itself, for a transport that moves it without packing: the
 * links (dir, parity) of the first (side 0) / last (side 1) owned slice (owned = 1) or of the
 * left (0) / right (1) halo slice (owned = 0) are `*pieces` runs of `*piece_bytes` bytes,
 * `*pitch_bytes` apart, the first at `*first`; run k is piece k of qcd_halo_pack's buffer. */
int    qcd_halo_region(qcd_ctx *ctx, int side, int owned, int dir, int parity, void **first,
                       size_t *piece_bytes, size_t *pitch_bytes, int *pieces);

/* UpdateEdges between two parts that live in ONE process (the reference's own single-process
 * BigLattice, suncl.cpp:4005-4061): boundary slice `pack_side` of src -> halo `recv_side` of dst,
 * same device or peer device; ordered after src's and before dst's later work. */
int    qcd_halo_copy(qcd_ctx *src, int pack_side, qcd_ctx *dst, int recv_side, int dir, int parity);

/* The BIGLAT branch of model::lattice_simulate (suncl.cpp:5026-5138): per (parity, direction) a
 * kernel_run(sun_update_*) for every part followed by model::UpdateEdges(dir) (suncl.cpp:4005-4061).
 * A ring holds the parts (contexts with halo = 1, in ring order) and issues such a pass in ONE call
 * without a host synchronisation.  A part whose pass runs as the wavefront-pair kernel updates its whole slab in
 * ONE launch that starts with the slices next to its two faces and writes a word of signal memory when those
 * are in memory (qcd_update_signal_edges); a stream beside the context's waits for that word and moves the
 * messages while the launch goes on with the interior.  Other parts (SU(2), gid mode, small or ragged slabs,
 * devices without hipStreamWaitValue32) take two launches on the context's stream, boundary slices first, and
 * the messages travel beside the second.
 *   QCD_RING_COPY: all parts in this process; messages by device-to-device / peer copies
 *                  (several parts may share a device: the reference's devParts, QCDGPU.cpp:129-134);
 *   QCD_RING_RCCL: ncclSend / ncclRecv between ring neighbours (xGMI between the GPUs of a node);
 *                  every part on a device of its own.  RCCL is bound at run time (dlopen).
 * qcd_ring_create: all parts driven by this process (the reference's one-process model; RCCL
 * communicators by ncclCommInitAll).  qcd_ring_create_rank: one part per process, `world` processes;
 * rank 0 makes an identifier with qcd_rccl_unique_id (128 bytes) and hands it to the others by its
 * own means (the reference has no multi-process mode: no counterpart).
 * qcd_ring_update / _overrelax: one pass + exchange; qcd_ring_sweep: n sweeps in the reference's
 * order; qcd_ring_fill_halos: every halo from its owner without an update, over the ring's transport (once
 * after a hot start or an upload: the reference's UpdateEdges x 4 behind its start kernels,
 * suncl.cpp:4835-4903).  Measurements stay per context (the *_sums entry points).
 * Destroy a ring BEFORE the contexts it was made of. */
enum { QCD_RING_COPY = 0, QCD_RING_RCCL = 1 };
typedef struct qcd_ring qcd_ring;
int  qcd_ring_create(qcd_ring **out, qcd_ctx *const *parts, int nparts, int transport);
int  qcd_rccl_unique_id(void *id, size_t bytes);
int  qcd_ring_create_rank(qcd_ring **out, qcd_ctx *part, int rank, int world, const void *id, size_t bytes);
void qcd_ring_destroy(qcd_ring *ring);
/* n sweeps the way the reference's CHB2 build runs its working cycles (suncl.cpp:4683-4729): every part runs the eight
 * passes of a sweep without an exchange, then the edges of all four directions are copied once.  Halos are up to a sweep
 * old inside a sweep: not the uncut lattice's chain at the slab faces, one round of messages per sweep instead of eight. */
int  qcd_ring_sweep_chb2(qcd_ring *ring, int n);
int  qcd_ring_update(qcd_ring *ring, int parity, int dir);
int  qcd_ring_overrelax(qcd_ring *ring, int parity, int dir);
int  qcd_ring_sweep(qcd_ring *ring, int n);
int  qcd_ring_fill_halos(qcd_ring *ring);
int  qcd_ring_synchronize(qcd_ring *ring);
/* how many passes of parts went out as one launch with the edge signal (qcd_update_signal_edges)
 * so far; the others took two launches */
int  qcd_ring_one_launch_passes(qcd_ring *ring, unsigned long long *n);

/* device_select / device_initialize (clinterface.cpp:345-459): what the host needs to know */
int qcd_device_count(int *count);
int qcd_device_name(int device, char *buf, size_t buflen);

/* GPU_debug.profiling + print_time_detailed (clinterface.cpp:1254-1266, 1626-1667) */
int qcd_set_profiling(qcd_ctx *ctx, int on);
int qcd_get_timing(qcd_ctx *ctx, qcd_timing *out);

/* runs n sweeps bracketed by HIP events on the context's stream; returns the elapsed device
 * time of the whole bracket in ms and the number of update launches inside it */
int qcd_timed_sweeps(qcd_ctx *ctx, int n, double *elapsed_ms, unsigned long long *launches);
/* the same for n overrelaxation sweeps (qcd_overrelax_sweep) */
int qcd_timed_overrelax_sweeps(qcd_ctx *ctx, int n, double *elapsed_ms, unsigned long long *launches);

/* number of random numbers consumed so far by all RANLUX streams of this context */
int qcd_rng_draws(qcd_ctx *ctx, unsigned long long *draws);

/* PRNG::check (random/random.cpp:339-402): the first `samples` numbers of RANLUX streams
 * 0 .. streams-1 produced on `device` against the host generator's, value by value;
 * *errors = how many differ (0 = the device generator reproduces the CPU one) */
int qcd_prng_check(int device, unsigned rand_series, int lux_level, int streams, int samples,
                   unsigned *errors);

/* self-check of the engine's own fp64 reciprocal, square root and logarithm (hardware seeds +
 * table; no counterpart in the reference, which calls the OpenCL built-ins): worst error, in
 * units of the last place, of log(1-x), sqrt(x), 1/(x+1e-3), 1/sqrt(x+1e-3) over n arguments
 * x = k * 2^-24 in (0, 1), against the host libm in long double */
int qcd_math_check(int device, int n, double max_ulp[4]);

/* the same for the accept step's fp64 cos(2 pi r) and sin(2 pi r) (a 256-entry table of angles plus a Taylor series for
 * the rest; the reference calls the OpenCL built-ins on the rounded product 2 pi r): worst ABSOLUTE error of the cosine
 * and of the sine over n arguments r of the forms the generator delivers, against the host libm in long double */
int qcd_trig_check(int device, int n, double max_abs_err[2]);

/* ---------------------------------------------------------------------------------------------
 * O(N) scalar path: the reference's second model (its O(N) tree, MODEL_ON = 1), one real field
 * value per site, checkerboard multi-hit Metropolis.  Same boundary as above: each entry point
 * replaces the kernel_run call sites of O(N)/suncl/suncl.cpp cited beside it. */

typedef struct qcd_on_ctx qcd_on_ctx;

/* run_parameters of the O(N) model (O(N)/suncl/suncl.h:74-102; init keys SKGROUP, ON_Z, ON_LAMBDA,
 * ON_ZETA, ON_ETA, ON_B, O(N)/suncl/suncl.cpp:273-286) */
typedef struct {
    int      L[4];         /* lattice_full_size: Nx, Ny, Nz, Nt.  Ny even (coord_work.cpp:151-175); odd Nx, Nz, Nt for a
                              context that owns the whole lattice (halo = 0) only */
    int      precision;    /* 0 = single, 1 = double (model_precision_single / _double) */
    int      nhit;         /* NHIT: proposals per site and pass */
    int      sk_group;     /* ON_SK_group: N of the O(N) group in the action */
    double   z, lambda, zeta, eta, b;   /* ON_z, ON_lambda, ON_zeta, ON_eta, ON_b */
    int      start_mode;   /* QCD_START_HOT or QCD_START_COLD */
    int      lux_level;    /* RANLUX luxury level 0..4 */
    unsigned rand_series;  /* RANDSERIES, non-zero */
    int      device;
} qcd_on_params;

/* model::model_lattice_init + lattice_create_buffers (O(N)/suncl/suncl.cpp:1241-1445, 1926-2090);
 * also derives ON_sqrt_z_lambda_zeta and ON_max_U (suncl.cpp:1244-1253) */
int  qcd_on_create(qcd_on_ctx **out, const qcd_on_params *params);
void qcd_on_destroy(qcd_on_ctx *ctx);

/* lattice_simulate_start (suncl.cpp:2158-2176; kernels lattice_init_cold / lattice_init_hot,
 * O(N)/oncl/oncl.cl:49-72): field = 0.25, or r * max_U; also resets the RANLUX streams */
int qcd_on_init_lattice(qcd_on_ctx *ctx);

/* kernel_run(sun_update_odd_X_id) / (sun_update_even_X_id) with the PRNG0->produce() in front of it
 * (suncl.cpp:2233-2256; kernels update_odd / update_even, oncl.cl:74-366); parity 0 even, 1 odd */
int qcd_on_update(qcd_on_ctx *ctx, int parity);
/* n times lattice_update: odd pass, then even pass (suncl.cpp:2228-2232) */
int qcd_on_sweep(qcd_on_ctx *ctx, int n);
/* the same bracketed by HIP events: device time of the n sweeps in ms */
int qcd_on_timed_sweeps(qcd_on_ctx *ctx, int n, double *elapsed_ms);

/* lattice_measure_action, lattice_measure_plq, lattice_measure_corr with their reduce kernels
 * (suncl.cpp:2177-2217; on_measurements_cl.cl:48-214), one pass over the field.  Per-site averages:
 * out[0] action S, out[1] physical field, out[2] its square, out[3] correlator with the site at
 * +step, out[4] correlator with the site at (+1,+1,+1,+1) */
int qcd_on_measure(qcd_on_ctx *ctx, const int step[4], double out[5]);

/* out[0]: random numbers consumed so far; out[1], out[2]: the acceptance sums of the last even and
 * odd pass (ACC_RATE, oncl.cl:86-94, 216-218: sum over sites of NHIT minus the proposals rejected
 * before the accepted one; the host divides by sites * NHIT, suncl.cpp:884) */
int qcd_on_counters(qcd_on_ctx *ctx, unsigned long long out[3]);
int qcd_on_max_u(const qcd_on_ctx *ctx, double *out);

/* buffer_map of lattice_table (suncl.cpp:2107, 2152): the field in the reference's site order,
 * element gid = y + z*Ny + t*Ny*Nz + x*Ny*Nz*Nt, float or double by precision */
int qcd_on_upload(qcd_on_ctx *ctx, const void *host_field, size_t bytes);
int qcd_on_download(qcd_on_ctx *ctx, void *host_field, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif
