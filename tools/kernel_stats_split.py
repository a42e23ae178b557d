#!/usr/bin/env python3
"""Per-direction averages of the update kernel from a rocprofv3 kernel trace of bench.py, with the launches of the
warm-up, of the timed brackets and of the legs behind them (the reference-chain comparison) reported apart:

    python3 tools/kernel_stats_split.py <..._kernel_trace.csv> <warmup sweeps> <steps> <brackets>

The timed launches are the ones whose durations bench.py's own HIP-event brackets add up; 8 x their mean is the
sweep time the bench line's ms_per_step should agree with."""
import csv
import json
import sys

path, warmup, steps, brackets = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        k = r["Kernel_Name"]
        if "k_update" in k and "qcd_f32" not in k:
            rows.append((int(r["Start_Timestamp"]), k, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3))
rows.sort()
cuts = [("warmup", 8 * warmup), ("timed", 8 * steps * brackets), ("after", len(rows))]
out, at = {}, 0
for name, n in cuts:
    part = rows[at:at + n] if name != "after" else rows[at:]
    at += len(part)
    per = {}
    for _, k, us in part:
        per.setdefault(k, []).append(us)
    out[name] = {"launches": len(part), "mean_us": (sum(us for _, _, us in part) / len(part)) if part else None,
                 "per_kernel_mean_us": {k: sum(v) / len(v) for k, v in sorted(per.items())}}
if out["timed"]["mean_us"]:
    out["timed"]["ms_per_sweep_from_kernel_durations"] = 8e-3 * out["timed"]["mean_us"]
    # bracket by bracket (bench.py reports the MEDIAN bracket: its ms_per_step is to be held against the median here)
    timed = rows[8 * warmup:8 * warmup + 8 * steps * brackets]
    per_bracket = [sum(us for _, _, us in timed[b * 8 * steps:(b + 1) * 8 * steps]) / (8 * steps) for b in range(brackets)]
    out["timed"]["mean_us_per_bracket"] = per_bracket
    out["timed"]["ms_per_sweep_per_bracket"] = [8e-3 * m for m in per_bracket]
    out["timed"]["ms_per_sweep_median_bracket"] = sorted(8e-3 * m for m in per_bracket)[brackets // 2]
print(json.dumps(out, indent=1))
