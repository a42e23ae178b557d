#!/usr/bin/env python3
"""bench.py -- link-updates/s of the SU(3) heat-bath sweep on MI355X (the headline metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one full sweep (8 passes: odd/even x X,Y,Z,T) of heat-bath + re-unitarisation over a
32^4 SU(3) fp64 lattice per GPU (weak scaling: N GPUs hold a (32N) x 32^3 lattice cut into x-slabs,
halos exchanged over RCCL after every pass).  Inputs are resident in HBM before the timed region.
Rank 0 prints ONE JSON line.

Started without a launcher (`python bench.py --gpus N`, WORLD_SIZE unset) with N > 1, the process
starts its N ranks itself as children (python -m torch.distributed.run) before it touches the GPU,
and passes their output through.

The timed region is `--brackets` brackets of exactly `--steps` sweeps, each between a barrier and
a device synchronisation on both sides; `value` and `ms_per_step` are the MEDIAN bracket, the
fastest and slowest are reported beside it.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BYTES_PER_LINK_UPDATE_SU3_F64 = 864     # 8 link reads + 1 link write, 12 reals of 8 B each
HBM_PEAK_GBS = 8000.0                   # MI355X HBM3E spec peak


def cpu_baseline(log):
    """Times the CPU path on this host: the reference's own binary (oracle/_ref, one thread, as
    the reference is single-threaded) when it travelled with the tree, else this repo's C port.
    Bounded sample: SU(3) 16^4, beta 6, NHIT 10, cold start, 32 sweeps (~15 s)."""
    ref = os.path.join(ROOT, "oracle", "_ref", "qcdgpu_cpu")
    sweeps, L = 32, 16
    sites = L ** 4
    out = None
    if os.path.exists(ref):
        try:
            def run(nav):
                tmp = tempfile.mkdtemp(prefix="qcdbench_")
                with open(os.path.join(tmp, "init.dat"), "w") as f:
                    f.write("GROUP = 3\nND = 4\nLS = %d\nLT = %d\nBETA = 6.0\nNHIT = 10\nINTS = 1\n"
                            "PRNG = RANLUX3\nRANDSERIES = 54\nNAV = %d\nITER = 1\nNITER = 1\n"
                            "OUTPUTPATH = %s/\n" % (L, L, nav, tmp))
                t0 = time.perf_counter()
                subprocess.run([ref, os.path.join(tmp, "init.dat")], check=True, cwd=tmp,
                               stdout=subprocess.DEVNULL, timeout=300)
                return time.perf_counter() - t0
            t_meas = run(0)            # initialisation + one measurement
            t_all = run(sweeps)        # the same + `sweeps` thermalisation sweeps
            dt = max(t_all - t_meas, 1e-9)
            out = {"value": 4.0 * sites * sweeps / dt, "unit": "link-updates/s", "cores": 1,
                   "kind": "reference",
                   "sample": "reference suncpp binary, SU(3) %d^4 beta=6.0 NHIT=10 cold RANLUX3, "
                             "%d sweeps, %.1f s" % (L, sweeps, dt)}
        except Exception as e:  # noqa: BLE001
            log("reference binary not usable here (%s); timing the C port instead" % e)
    if out is None:
        cli = os.path.join(ROOT, "oracle", "qcd_oracle_cli")
        if not os.path.exists(cli):
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "qcd_oracle_cli"], check=True,
                           stdout=subprocess.DEVNULL)
        tmp = tempfile.mkdtemp(prefix="qcdbench_")
        with open(os.path.join(tmp, "init.dat"), "w") as f:
            f.write("GROUP = 3\nND = 4\nLS = %d\nLT = %d\nBETA = 6.0\nNHIT = 10\nINTS = 1\n"
                    "PRNG = RANLUX3\nRANDSERIES = 54\n" % (L, L))
        r = subprocess.run([cli, os.path.join(tmp, "init.dat"), "--time-sweeps", str(sweeps)],
                           check=True, capture_output=True, text=True, timeout=300)
        j = json.loads(r.stdout.strip().splitlines()[-1])
        out = {"value": j["link_updates_per_s"], "unit": "link-updates/s", "cores": 1, "kind": "port",
               "sample": "C restatement of suncpp, SU(3) %d^4 beta=6.0 NHIT=10 cold RANLUX3, %d sweeps, "
                         "%.1f s" % (L, j["sweeps"], j["seconds"])}
    return out


def measured_traffic(key="hbm_bytes_per_update_launch"):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/), or None."""
    p = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    if os.path.exists(p):
        try:
            with open(p) as f:
                return json.load(f).get(key)
        except Exception:  # noqa: BLE001
            return None
    return None


def checker_fixture(L, beta, share):
    """Plaquette after sweeps 1 and 2 of the CPU checker's per-thread-stream mode at this size, `share` sites of one
    parity per stream (tests/golden/, minted by oracle/make_golden_persite32.py), or None."""
    if tuple(L) != (32, 32, 32, 32) or beta != 6.0:
        return None
    p = os.path.join(ROOT, "tests", "golden", "persite_su3_32x32x32x32_b6%s.json" % ("_share%d" % share if share > 1 else ""))
    try:
        with open(p) as f:
            return json.load(f)
    except Exception:  # noqa: BLE001
        return None


def reference_chain(L, beta):
    """Mean and binned error of Plq_total over the thermalised samples of the reference program's chain at this size
    (tests/golden/stat_*.json, minted by oracle/make_golden_stat.py), or None."""
    if len(set(L)) != 1 or beta != 6.0:
        return None
    name = "stat_su3_%dx%dx%dx%d_b6" % tuple(L)
    try:
        with open(os.path.join(ROOT, "tests", "golden", name + ".json")) as f:
            doc = json.load(f)
        x = doc["plq_total"][:500]
        nav = int(doc["init"]["NAV"])
    except Exception:  # noqa: BLE001
        return None
    nb = 10
    per = len(x) // nb
    if per < 2:
        return None
    bins = [sum(x[i * per:(i + 1) * per]) / per for i in range(nb)]
    m = sum(x[:nb * per]) / (nb * per)
    err = (sum((b - m) ** 2 for b in bins) / (nb - 1) / nb) ** 0.5
    return {"name": name, "mean": m, "err": err, "n": nb * per, "nav": nav}


def launch_ranks(n):
    """WORLD_SIZE unset and --gpus n > 1: start the n ranks as children of this process (which has
    not touched the GPU), pass rank 0's JSON line through and return the launcher's exit code."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)   # the first sweeps from the cold start run faster than thermalised ones
    ap.add_argument("--L", type=int, default=32, help="lattice extent per GPU (Nx per GPU, Ny, Nz, Nt)")
    ap.add_argument("--beta", type=float, default=6.0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--brackets", type=int, default=5, help="timed brackets of --steps sweeps each (median reported)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="at N > 1: weak = an L^4 lattice per GPU (lattice (N L) x L^3, the default); strong = ONE L^4 lattice cut "
                         "into N slabs of L/N slices (the reference's N1 = {L/N, ...}, SU(N)/init_2.dat:10)")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    def log(msg):
        if rank == 0:
            print("[bench] " + msg, file=sys.stderr, flush=True)

    # a run that has not finished after this many seconds (a halo exchange that never completes, say) ends itself
    # with a message instead of holding its GPUs until somebody else's limit strikes
    import threading

    def _give_up():
        print("[bench] rank %d: no result after %s s, giving up" % (rank, os.environ.get("QCDGPU_BENCH_WATCHDOG", "600")),
              file=sys.stderr, flush=True)
        os._exit(3)

    watchdog = threading.Timer(float(os.environ.get("QCDGPU_BENCH_WATCHDOG", "600")), _give_up)
    watchdog.daemon = True
    watchdog.start()

    import torch
    import qcdgpu_amd as q   # raises if the HIP library is missing: no fallback
    from qcdgpu_amd import biglattice as bl

    if not torch.cuda.is_available():
        sys.exit("no GPU visible")
    # rehearsal switches for a one-GPU box (never set by the driver): all ranks on device 0,
    # halos over gloo through host memory instead of RCCL
    if os.environ.get("QCDGPU_BENCH_SHARE_DEVICE"):
        local_rank = 0
    backend = os.environ.get("QCDGPU_BENCH_BACKEND", "nccl")
    # how the halos travel at N > 1: "ring" = the C library's own pass (qcd_ring: RCCL send/receive
    # issued below the C-ABI; torch.distributed over gloo only hands the RCCL identifier round and
    # adds up timings and plaquette sums), "torch" = the Python driver over torch.distributed P2P
    transport = os.environ.get("QCDGPU_BENCH_TRANSPORT", "torch" if os.environ.get("QCDGPU_BENCH_SHARE_DEVICE") else "ring")
    if transport == "ring" and world > 1:
        backend = "gloo"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    p2p_group = None
    n = args.L
    strong = args.scaling == "strong" and world > 1
    if strong and n % world:
        sys.exit("--scaling strong: %d GPUs do not divide L = %d" % (world, n))
    L = (n, n, n, n) if strong else (n * world, n, n, n)
    sites_per_gpu = n ** 4 // world if strong else n ** 4
    kw = dict(nhit=10, start_mode=q.START_COLD, lux_level=3, rand_series=54)
    if strong:
        # Sites per RANLUX stream (qcd_params.rng_share).  The library chooses it from the full lattice, assuming one
        # device holds it all; a thin slab has too few groups of sites to keep a device busy with shared streams, so
        # the cut decides here.  Every rank passes the same value (the stream map is a function of global coordinates).
        kw["rng_share"] = q.rng_share_for(L, n // world)

    if world == 1:
        ctx = q.Context(group=3, L=L, beta=args.beta, device=local_rank, **kw)
        ctx.init_lattice()
        sweep = ctx.sweep
        plaquette = ctx.plaquette
        sync = ctx.synchronize
    else:
        lat = None
        if transport == "ring":
            # every rank must end up on the same transport: agree on whether the ring came up everywhere
            try:
                lat = bl.make_rank_ring(3, L, args.beta, rank, world, local_rank, **kw)
                problem = None
            except Exception as e:  # noqa: BLE001
                problem = e
            ok = torch.tensor([0 if problem is not None else 1], dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                if problem is not None:
                    print("[bench] rank %d: no RCCL ring (%s)" % (rank, problem), file=sys.stderr, flush=True)
                log("the C library's RCCL ring did not come up on every rank: halos over torch.distributed P2P (RCCL) instead")
                if lat is not None:
                    lat.close()
                    lat.ctx.close()
                lat = None
                transport = "torch"
                if os.environ.get("QCDGPU_BENCH_BACKEND") == "gloo":   # rehearsal on one GPU: staged through the host
                    backend = "gloo"
                else:
                    p2p_group = dist.new_group(backend="nccl", device_id=torch.device("cuda", local_rank))
                    backend = "nccl (second process group)"
        if lat is None:
            lat = bl.make_rank_slab(3, L, args.beta, rank, world, local_rank, dist_group=p2p_group, **kw)
        ctx = lat.engines[0].ctx
        sweep = lat.sweep
        plaquette = lat.plaquette

        def sync():
            torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()
        sync()
        torch.cuda.synchronize()

    def global_int(v):
        if dist is None:
            return int(v)
        t = torch.tensor([int(v)], dtype=torch.int64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return int(t.item())

    def global_draws():
        return global_int(ctx.rng_draws())

    # the first two sweeps from the cold start are compared with the CPU checker's numbers for the
    # same lattice, seed and stream layout (a committed fixture; they are part of the warm-up).  A cut
    # lattice is compared like an uncut one: results do not depend on the cut.
    fixture = checker_fixture(L, args.beta, ctx.rng_share)
    parity_check = None
    done = 0
    if fixture is not None and args.warmup >= 2:
        worst, draws_equal = 0.0, True
        for row in fixture["sweeps"]:
            sweep(1)
            done += 1
            got = plaquette()
            worst = max(worst, max(abs(g - w) / abs(w) for g, w in zip(got, row["plaquette"])))
            draws_equal = draws_equal and global_draws() == int(row["rng_draws"])
        parity_check = {"against": "CPU checker, per-thread streams, %s" % fixture["name"],
                        "sweeps": len(fixture["sweeps"]), "max_rel_diff_plaquette": worst,
                        "rng_draws_equal": draws_equal, "ok": bool(worst < 1e-12 and draws_equal)}
        log("parity at 32^4 after %d sweeps: max rel diff %.2e, draws equal %s" % (done, worst, draws_equal))
    elif (world > 1 and (transport == "ring" or os.environ.get("QCDGPU_BENCH_FORCE_CROSSCHECK")) and args.warmup >= 2
          and not os.environ.get("QCDGPU_BENCH_NO_CROSSCHECK")):
        # No CPU fixture at this size: the same slabs driven by the second implementation of the exchange (the Python
        # driver over the process group that is there anyway: host-staged when that is gloo) must give the same
        # plaquette and draw count after the first two sweeps; a ring that comes up and moves wrong halos shows here.
        # (QCDGPU_BENCH_FORCE_CROSSCHECK: run it whatever the transport, to rehearse this code on one GPU.)
        try:
            other = bl.make_rank_slab(3, L, args.beta, rank, world, local_rank, dist_group=p2p_group, **kw)
            worst, draws_equal = 0.0, True
            for _ in range(2):
                sweep(1)
                other.sweep(1)
                done += 1
                got, want = plaquette(), other.plaquette()
                worst = max(worst, max(abs(g - w) / abs(w) for g, w in zip(got, want)))
                draws_equal = draws_equal and global_draws() == global_int(other.engines[0].ctx.rng_draws())
            other.engines[0].ctx.close()
            parity_check = {"against": "the same slabs with the halos exchanged by the Python driver over torch.distributed",
                            "sweeps": 2, "max_rel_diff_plaquette": worst, "rng_draws_equal": draws_equal,
                            "ok": bool(worst < 1e-12 and draws_equal)}
            log("cross-check against the second exchange after 2 sweeps: max rel diff %.2e, draws equal %s" % (worst, draws_equal))
        except Exception as e:  # noqa: BLE001  (the check must not take the measurement down with it)
            parity_check = {"against": "the second exchange", "error": str(e)[:300], "ok": False}
            print("[bench] rank %d: cross-check failed to run: %s" % (rank, e), file=sys.stderr, flush=True)
    log("warm-up: %d sweeps of %dx%dx%dx%d SU(3) fp64 on %d GPU(s)" % ((args.warmup,) + L + (world,)))
    sweep(max(args.warmup - done, 0))
    barrier()
    launches0 = ctx.timing().update_launches
    walls, evs = [], []
    for _ in range(max(args.brackets, 1)):
        barrier()
        t0 = time.perf_counter()
        if world == 1:
            ev_ms, _ = ctx.timed_sweeps(args.steps)      # HIP events on the library's own stream
        else:
            sweep(args.steps)
            ev_ms = None
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            ev_ms = 1e3 * dt        # no single stream carries a slab pass: the wall clock between the barriers
        walls.append(dt)
        evs.append(ev_ms)
    order = sorted(range(len(walls)), key=lambda i: walls[i])
    mid = order[len(order) // 2]
    dt, ev_ms = walls[mid], evs[mid]
    launches = (ctx.timing().update_launches - launches0) // len(walls)
    plq = plaquette()
    sweeps_done = args.warmup + args.steps * len(walls)

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    if parity_check is not None and not parity_check["ok"]:
        log("PARITY CHECK FAILED: %s" % json.dumps(parity_check))   # stays in the line as mean_plaquette.parity_check.ok = false
    link_updates = 4.0 * sites_per_gpu * world * args.steps
    value = link_updates / dt
    # dominant kernel: k_update, V/2 link updates per launch; average launch duration from the
    # device-side event bracket around the timed sweeps (it contains nothing but those launches
    # on one GPU; with slabs it also contains the halo traffic, so the fraction is conservative)
    # (a slab updates a pass in two launches, boundary slices first: count passes, not launches)
    passes = 8 * args.steps
    assert launches >= passes, (launches, passes)
    # which kernel a pass over this context's slices runs as (the pair kernel needs enough blocks of sites)
    kernel_name = "k_update_gh" if ctx.pass_uses_pair_kernel() else "k_update"
    if world > 1 and launches > passes:
        kernel_name += " (boundary slices and interior as two launches)"
    per_launch_s = (ev_ms * 1e-3) / passes
    bytes_per_launch = BYTES_PER_LINK_UPDATE_SU3_F64 * sites_per_gpu / 2
    achieved = bytes_per_launch / per_launch_s / 1e9
    result = {
        "metric": "link-updates/sec + mean plaquette, 32^4 SU(3) at 1/2/4/8 GPUs",
        "value": value,
        "unit": "link-updates/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic (cold start, beta=%.2f Markov chain, RANLUX3 per-thread streams, seed 54)" % args.beta,
        "config": {"workload": ("SU(3) %dx%dx%dx%d heat-bath sweep (NHIT=10, re-unitarised), " % L)
                               + ("one lattice cut into %d slabs of %d slices" % (world, n // world) if strong else "%d^4 per GPU" % n),
                   "lattice": list(L), "beta": args.beta, "nhit": 10,
                   "prng": "RANLUX3", "rng_share": ctx.rng_share, "parallelism": "x-slabs x%d" % world,
                   "halo_transport": None if world == 1 else ("RCCL send/recv issued by the C library (qcd_ring)" if transport == "ring"
                                                              else "torch.distributed P2P, backend %s" % backend)},
        "mean_plaquette": {"Plq_spat": plq[0], "Plq_temp": plq[1], "Plq_total": plq[2],
                           "after_sweeps": sweeps_done, "parity_check": parity_check},
        "brackets": {"n": len(walls), "steps_each": args.steps, "value_median": value,
                     "value_min": link_updates / max(walls), "value_max": link_updates / min(walls),
                     "spread": (max(walls) - min(walls)) / dt},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     # PMC traffic was collected on this workload (32^4 SU(3) per GPU) only
                     "traffic": measured_traffic() if (n == 32 and args.beta == 6.0) else None,
                     "kernel": kernel_name, "launches": int(launches),
                     "avg_launch_us": 1e6 * per_launch_s,
                     "algorithmic_bytes_per_launch": bytes_per_launch},
    }
    if world == 1:
        # the other half of the link update the workload names: overrelaxation sweeps on the thermalised configuration,
        # device time per pass from a HIP-event bracket like the headline's (not part of `value`; same algorithmic bytes)
        try:
            ctx.overrelax_sweep(2)
            o_ms, o_launches = ctx.timed_overrelax_sweeps(10)
            o_pass_s = 1e-3 * o_ms / o_launches
            o_gbs = bytes_per_launch / o_pass_s / 1e9
            result.setdefault("other_workloads", {})["su3_%dx4_overrelaxation" % n] = {
                "link_updates_per_s": (sites_per_gpu / 2) / o_pass_s, "ms_per_sweep": 8e3 * o_pass_s,
                "workload": "SU(3) %d^4 overrelaxation sweep (microcanonical reflection in the three SU(2) subgroups, re-unitarised), fp64" % n,
                "roofline": {"bound": "hbm", "achieved": o_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": o_gbs / HBM_PEAK_GBS,
                             "kernel": "k_update_gh<.., OVR>" if ctx.pass_uses_pair_kernel() else "k_overrelax",
                             "avg_launch_us": 1e6 * o_pass_s, "algorithmic_bytes_per_launch": bytes_per_launch,
                             "traffic": measured_traffic("overrelaxation_hbm_bytes_per_launch") if (n == 32 and args.beta == 6.0) else None}}
        except q.QcdError as e:
            log("overrelaxation leg skipped: %s" % e)
        # the chain's mean plaquette beside the mean of the reference program's own chain at this size (a committed fixture:
        # different random streams, the same Markov kernel and start), outside the timed region
        chain = reference_chain(L, args.beta)
        if chain is not None:
            # the same window of the chain as the fixture's: cold start, the fixture's thermalisation sweeps, then one
            # measurement per sweep (from a cold start the plaquette is still settling after hundreds of sweeps, so the two
            # means are compared over the same sweeps, not over any thermalised stretch)
            with q.Context(group=3, L=L, beta=args.beta, device=local_rank, **kw) as c2:
                c2.init_lattice()
                c2.sweep(chain["nav"])
                mine = []
                for _ in range(chain["n"]):
                    c2.sweep(1)
                    mine.append(c2.plaquette()[2])
            per = len(mine) // 10
            bins = [sum(mine[i * per:(i + 1) * per]) / per for i in range(10)]
            m_gpu = sum(mine[:10 * per]) / (10 * per)
            e_gpu = (sum((b - m_gpu) ** 2 for b in bins) / 9.0 / 10.0) ** 0.5
            sigma = (e_gpu ** 2 + chain["err"] ** 2) ** 0.5
            result["mean_plaquette"]["reference_chain"] = {
                "fixture": chain["name"], "sweeps": "%d..%d from the cold start, one sample per sweep" % (chain["nav"] + 1, chain["nav"] + chain["n"]),
                "reference_mean": chain["mean"], "reference_err": chain["err"], "gpu_mean": m_gpu, "gpu_err": e_gpu,
                "difference_in_sigma": abs(m_gpu - chain["mean"]) / sigma, "ok": bool(abs(m_gpu - chain["mean"]) < 4.0 * sigma)}
            log("mean plaquette over sweeps %d..%d: %.6f +- %.6f, reference chain %.6f +- %.6f"
                % (chain["nav"] + 1, chain["nav"] + chain["n"], m_gpu, e_gpu, chain["mean"], chain["err"]))
    if world == 1 and not args.no_cpu:
        # the scalar-model configuration beside the headline workload (not part of `value`): the O(N)
        # scalar model, 64^4 in single precision, NHIT = 10, site updates per second from HIP events
        try:
            with q.OnContext((64, 64, 64, 64), precision=0, nhit=10, rand_series=7, device=local_rank) as oc:
                oc.init_lattice()
                oc.sweep(5)
                on_ms = oc.timed_sweeps(20)
                # algorithmic bytes per site update: 9 field values read and one written, 4 B each; a
                # pass updates half the sites.  The kernel is bound by vector-instruction issue
                # (logarithms, exponentials and the generator), not by memory: the HBM fraction says
                # how far it is from the other roof, `traffic` what it really moves (generator state)
                on_pass_s = on_ms * 1e-3 / 40
                on_bytes = 40.0 * 64.0 ** 4 / 2
                on_gbs = on_bytes / on_pass_s / 1e9
                result.setdefault("other_workloads", {})["on_sigma_64x4_f32"] = {
                    "site_updates_per_s": 64.0 ** 4 * 20 / (on_ms * 1e-3), "ms_per_sweep": on_ms / 20,
                    "workload": "O(N) scalar model 64^4, single precision, NHIT=10, RANLUX3, cold start, 20 sweeps",
                    "roofline": {"bound": "hbm", "achieved": on_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": on_gbs / HBM_PEAK_GBS, "kernel": "k_on_update<float>",
                                 "avg_launch_us": 1e6 * on_pass_s, "algorithmic_bytes_per_launch": on_bytes,
                                 "traffic": measured_traffic("on_hbm_bytes_per_update_launch"),
                                 "limited_by": measured_traffic("on_limited_by")}}
        except q.QcdError as e:
            log("O(N) leg skipped: %s" % e)
        # the headline lattice in single precision (the reference's PRECISION switch; its paper quotes every
        # timing in both precisions): the float build of the library, not part of `value`
        try:
            with q.Context(group=3, L=L, beta=args.beta, nhit=10, device=local_rank, precision=1) as c32:
                c32.init_lattice()
                c32.sweep(max(args.warmup, 20))   # thermalised enough for steady accept rates (the headline leg has its brackets for that)
                f_ms, f_launches = c32.timed_sweeps(args.steps)
                f_pass_s = f_ms * 1e-3 / f_launches
                f_bytes = 432.0 * (n ** 4) / 2      # 8 links read + 1 written, 48 B each, per updated link
                f_gbs = f_bytes / f_pass_s / 1e9
                result.setdefault("other_workloads", {})["su3_%dx4_f32" % n] = {
                    "link_updates_per_s": 4.0 * n ** 4 * args.steps / (f_ms * 1e-3), "ms_per_sweep": f_ms / args.steps,
                    "workload": "SU(3) %d^4 heat-bath sweep in single precision (float links and arithmetic, fp64 sums), beta %.2f" % (n, args.beta),
                    "mean_plaquette": float(c32.plaquette()[2]),
                    "roofline": {"bound": "hbm", "achieved": f_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": f_gbs / HBM_PEAK_GBS, "kernel": "qcd_f32::k_update_gh", "avg_launch_us": 1e6 * f_pass_s,
                                 "algorithmic_bytes_per_launch": f_bytes,
                                 "traffic": measured_traffic("f32_hbm_bytes_per_update_launch")}}
        except (q.QcdError, ImportError) as e:
            log("single-precision leg skipped: %s" % e)
        log("CPU baseline leg (bounded sample, one core)")
        result["cpu_baseline"] = cpu_baseline(log)
    else:
        result["cpu_baseline"] = None
    print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
