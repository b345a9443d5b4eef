#!/usr/bin/env python3
"""Average a rocprofv3 --pmc counter per kernel name from *counter_collection.csv; when the pass also left a kernel trace, the
average dispatch duration of each kernel is added as the pseudo-counter DURATION_NS (what MFMA-busy cycles are divided by)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    d, out = sys.argv[1], sys.argv[2]
    agg = defaultdict(lambda: defaultdict(list))
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(fn) as f:
            for r in csv.DictReader(f):
                name = (r.get("Kernel_Name") or "").split("(")[0][-48:]
                cname = r.get("Counter_Name") or ""
                try:
                    val = float(r.get("Counter_Value") or 0)
                except ValueError:
                    continue
                agg[name][cname].append(val)
    for fn in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(fn) as f:
            for r in csv.DictReader(f):
                name = (r.get("Kernel_Name") or "").split("(")[0][-48:]
                try:
                    agg[name]["DURATION_NS"].append(float(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
                except (KeyError, ValueError):
                    continue
    res = {k: {c: {"calls": len(v), "avg": sum(v) / len(v), "min": min(v), "max": max(v)} for c, v in cs.items()}
           for k, cs in agg.items()}
    with open(out, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print("wrote", out, len(res), "kernels")


if __name__ == "__main__":
    main()
