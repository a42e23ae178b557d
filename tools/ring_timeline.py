#!/usr/bin/env python3
"""Kernel timeline of the last passes in a rocprofv3 kernel trace of tools/ring_trace.py:
usage: ring_timeline.py <kernel_trace.csv> [number of kernels from the end]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = r["Kernel_Name"]
    name = name[name.find("k_"):][:40] if "k_" in name else name[:40]
    print("%9.1f -> %9.1f us  queue %s  grid %8s wg %4s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3,
                                                                r.get("Queue_Id", "?"), r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), name))
