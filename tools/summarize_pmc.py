#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc CSV output (counter_collection files)."""
import collections
import csv
import glob
import json
import os
import sys


def main(outdir):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"].split("(")[0]
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    res = {}
    for k, cs in acc.items():
        res[k] = {c: {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)} for c, v in cs.items()}
    json.dump(res, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1])
