#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats output directory into a small per-kernel table
(calls, total/avg/min/max duration) suitable for committing under profiles/."""
import csv
import glob
import json
import os
import sys


def main():
    d, out = sys.argv[1], sys.argv[2]
    rows = []
    files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if files:
        with open(files[0]) as f:
            for r in csv.DictReader(f):
                rows.append(r)
        keys = rows[0].keys() if rows else []
        with open(out, "w") as f:
            f.write("# source: %s\n" % os.path.basename(files[0]))
            f.write(",".join(keys) + "\n")
            for r in rows:
                f.write(",".join('"%s"' % r[k] if "," in r[k] else r[k] for k in keys) + "\n")
        print("wrote", out, len(rows), "kernels")
        return
    # fall back to aggregating the raw trace
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    agg = {}
    for fn in files:
        with open(fn) as f:
            for r in csv.DictReader(f):
                name = r.get("Kernel_Name") or r.get("kernel_name")
                dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                a = agg.setdefault(name, [0, 0, 1 << 62, 0])
                a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    with open(out, "w") as f:
        f.write("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs\n")
        for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write('"%s",%d,%d,%.1f,%d,%d\n' % (name, a[0], a[1], a[1] / a[0], a[2], a[3]))
    print("wrote", out, len(agg), "kernels (aggregated from trace)")


if __name__ == "__main__":
    main()
