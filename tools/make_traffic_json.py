#!/usr/bin/env python3
"""profiles/roofline_traffic.json from a tools/collect_profiles.sh output directory: HBM bytes per
k_update launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB counters; the factor 2 on the read side is the
gfx950 correction, checked in the same run on k_reunit, whose traffic is known exactly).

usage: python3 tools/make_traffic_json.py gpurun_out/<dir> profiles/r01"""
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
pmc = json.load(open(os.path.join(src, "pmc_summary.json")))
per, per32, per_ovr = {}, {}, {}
for k, v in pmc.items():
    if "k_update" in k and "redo" not in k:
        ovr = k.replace(" ", "").endswith("true>")   # the overrelaxation pass through the same kernel (bench.py's extra leg)
        (per_ovr if ovr else per32 if "qcd_f32::" in k else per)[k] = {"fetch_bytes": 2 * 1024.0 * v["FETCH_SIZE"]["mean_per_dispatch"],
                                                                        "write_bytes": 1024.0 * v["WRITE_SIZE"]["mean_per_dispatch"]}
fetch = sum(p["fetch_bytes"] for p in per.values()) / len(per)
write = sum(p["write_bytes"] for p in per.values()) / len(per)
cal_f = json.load(open(os.path.join(src, "calib_fetch.json")))
cal_w = json.load(open(os.path.join(src, "calib_write.json")))
kf = [v for k, v in cal_f.items() if "qcd::k_reunit" in k][0]["FETCH_SIZE"]["mean_per_dispatch"]
kw = [v for k, v in cal_w.items() if "qcd::k_reunit" in k][0]["WRITE_SIZE"]["mean_per_dispatch"]
true_bytes = 402653184.0
out = {
    "hbm_bytes_per_update_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write, "per_kernel": per,
    "source": "%s/pmc_summary.json (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, bench.py --steps 3 "
              "--warmup 1, 32^4 SU(3))" % dst,
    "correction": "counters are in KiB; FETCH_SIZE doubled (gfx950 reports half the bytes of 16 B/lane streaming reads), WRITE_SIZE as is",
    "calibration": {"kernel": "k_reunit 32^4 SU(3): reads %d B and writes %d B per launch" % (true_bytes, true_bytes),
                    "FETCH_SIZE_KiB": kf, "WRITE_SIZE_KiB": kw,
                    "fetch_ratio_reported_over_true": 1024.0 * kf / true_bytes,
                    "write_ratio_reported_over_true": 1024.0 * kw / true_bytes},
}
if per_ovr:
    out["overrelaxation_hbm_bytes_per_launch"] = sum(p["fetch_bytes"] + p["write_bytes"] for p in per_ovr.values()) / len(per_ovr)
    out["overrelaxation_per_kernel"] = per_ovr
# the single-precision library's update pass, when tools/profile_f32.sh ran into <src>/f32
f32_dir = os.path.join(src, "f32")
if os.path.exists(os.path.join(f32_dir, "fetch.json")):
    f = json.load(open(os.path.join(f32_dir, "fetch.json")))
    w = json.load(open(os.path.join(f32_dir, "write.json")))
    keys = [k for k in f if "qcd_f32::k_update_gh" in k]
    fb = sum(2 * 1024.0 * f[k]["FETCH_SIZE"]["mean_per_dispatch"] for k in keys) / len(keys)
    wb = sum(1024.0 * w[k]["WRITE_SIZE"]["mean_per_dispatch"] for k in keys) / len(keys)
    out["f32_hbm_bytes_per_update_launch"] = fb + wb
    out["f32_fetch_bytes"], out["f32_write_bytes"] = fb, wb
    kf32 = [v for k, v in cal_f.items() if "qcd_f32::k_reunit" in k]
    kw32 = [v for k, v in cal_w.items() if "qcd_f32::k_reunit" in k]
    if kf32 and kw32:
        out["f32_calibration"] = {"kernel": "qcd_f32::k_reunit 32^4 SU(3): reads %d B and writes %d B per launch (8 B per lane)" % (true_bytes / 2, true_bytes / 2),
                                  "fetch_ratio_reported_over_true": 1024.0 * kf32[0]["FETCH_SIZE"]["mean_per_dispatch"] / (true_bytes / 2),
                                  "write_ratio_reported_over_true": 1024.0 * kw32[0]["WRITE_SIZE"]["mean_per_dispatch"] / (true_bytes / 2)}
    os.makedirs(dst, exist_ok=True)
    for name in ("fetch", "write", "insts", "cache"):
        if os.path.exists(os.path.join(f32_dir, name + ".json")):
            shutil.copy(os.path.join(f32_dir, name + ".json"), os.path.join(dst, "f32_pmc_%s.json" % name))
    for fn in os.listdir(os.path.join(f32_dir, "trace")):
        if fn.endswith("kernel_stats.csv"):
            shutil.copy(os.path.join(f32_dir, "trace", fn), os.path.join(dst, "f32_kernel_stats.csv"))
# the O(N) update pass, when tools/profile_on.sh ran into <src>/on
on_dir = os.path.join(src, "on")
if os.path.exists(os.path.join(on_dir, "fetch.json")):
    f = json.load(open(os.path.join(on_dir, "fetch.json")))
    w = json.load(open(os.path.join(on_dir, "write.json")))
    i = json.load(open(os.path.join(on_dir, "insts.json")))
    key = [k for k in f if "k_on_update<float" in k][0]
    fb = 2 * 1024.0 * f[key]["FETCH_SIZE"]["mean_per_dispatch"]
    wb = 1024.0 * w[key]["WRITE_SIZE"]["mean_per_dispatch"]
    out["on_hbm_bytes_per_update_launch"] = fb + wb
    out["on_fetch_bytes"], out["on_write_bytes"] = fb, wb
    valu = i[key]["SQ_INSTS_VALU"]["mean_per_dispatch"]
    out["on_valu_wave_instructions_per_launch"] = valu
    out["on_limited_by"] = ("vector-instruction issue: %.0f M wavefront instructions per pass (logarithms, exponential, generator), "
                            "one per SIMD every %.1f cycles of the 2.4 GHz nominal clock at the measured pass time; the bytes moved are "
                            "generator state, not field values" % (valu / 1e6, 0))
    for name in ("fetch", "write", "insts", "busy"):
        if os.path.exists(os.path.join(on_dir, name + ".json")):
            os.makedirs(dst, exist_ok=True)
            shutil.copy(os.path.join(on_dir, name + ".json"), os.path.join(dst, "on64_pmc_%s.json" % name))
    for fn in os.listdir(os.path.join(on_dir, "trace")) if os.path.isdir(os.path.join(on_dir, "trace")) else []:
        if fn.endswith("kernel_stats.csv"):
            shutil.copy(os.path.join(on_dir, "trace", fn), os.path.join(dst, "on64_kernel_stats.csv"))
            import csv
            for row in csv.DictReader(open(os.path.join(on_dir, "trace", fn))):
                if "k_on_update<float" in row["Name"]:
                    avg_ns = float(row["AverageNs"])
                    cyc = avg_ns * 2.4 * 1024 / valu
                    out["on_limited_by"] = ("vector-instruction issue: %.0f M wavefront instructions per pass (logarithms, exponential, "
                                            "generator) in %.0f us = one per SIMD every %.1f cycles of the 2.4 GHz nominal clock; the bytes "
                                            "moved are generator state, not field values" % (valu / 1e6, avg_ns / 1e3, cyc))
json.dump(out, open(os.path.join(os.path.dirname(dst.rstrip("/")), "roofline_traffic.json"), "w"), indent=1)
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "pmc_summary.json"), os.path.join(dst, "pmc_summary.json"))
shutil.copy(os.path.join(src, "calib_fetch.json"), os.path.join(dst, "calib_reunit_fetch.json"))
shutil.copy(os.path.join(src, "calib_write.json"), os.path.join(dst, "calib_reunit_write.json"))
shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(dst, "bench_under_rocprof.json"))
for f in os.listdir(os.path.join(src, "stats")):
    if f.endswith("kernel_stats.csv"):
        shutil.copy(os.path.join(src, "stats", f), os.path.join(dst, "kernel_stats_bench_32x4.csv"))
print(json.dumps({k: out[k] for k in ("hbm_bytes_per_update_launch", "fetch_bytes", "write_bytes")}), out["calibration"])
