#!/usr/bin/env python3
"""Where the time BETWEEN the kernels of a request's prompt pass goes: from a rocprofv3 --kernel-trace of `tools/ttft_ab.py --reps N 0`
take the LAST request (from its first patch-gather kernel to the last kernel of the trace), and print the wall time first start -> last end,
the sum of the kernel durations, the sum of the gaps, and the gaps grouped by the kernel in FRONT of them.
    python tools/ttft_gaps.py <rocprof output dir>"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    rows = []
    for fn in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(fn) as f:
            for r in csv.DictReader(f):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0][-48:]))
    rows.sort()
    first = [i for i, r in enumerate(rows) if "im2col" in r[2]]
    if not first:
        print("no im2col kernel in the trace"); return
    req = rows[first[-1]:]
    wall = (req[-1][1] - req[0][0]) / 1e3
    busy = sum(e - s for s, e, _ in req) / 1e3
    gaps = defaultdict(list)
    for (s0, e0, n0), (s1, e1, n1) in zip(req, req[1:]):
        gaps[n0].append((s1 - e0) / 1e3)
    total_gap = sum(sum(v) for v in gaps.values())
    print(f"last request: {len(req)} launches, first start -> last end {wall:.1f} us, kernel durations {busy:.1f} us, gaps {total_gap:.1f} us "
          f"({total_gap / max(len(req) - 1, 1):.2f} us per boundary)")
    print(f"{'kernel in front of the gap':50s} {'n':>5s} {'sum us':>9s} {'avg':>7s} {'min':>7s} {'max':>7s}")
    for n, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
        print(f"{n:50s} {len(v):5d} {sum(v):9.1f} {sum(v) / len(v):7.2f} {min(v):7.2f} {max(v):7.2f}")
    durs = defaultdict(list)
    for s, e, n in req:
        durs[n].append((e - s) / 1e3)
    print(f"\n{'kernel':50s} {'n':>5s} {'sum us':>9s} {'avg':>7s}")
    for n, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
        print(f"{n:50s} {len(v):5d} {sum(v):9.1f} {sum(v) / len(v):7.2f}")


if __name__ == "__main__":
    main()
