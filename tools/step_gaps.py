#!/usr/bin/env python3
"""Gaps between the kernels of the decode loop, from a rocprofv3 --kernel-trace of bench.py: inside a step (kernel -> next kernel) and ACROSS the step boundary
(finish_step_kernel of step i -> the first kernel of step i + 1 = one hipGraphLaunch to the next).    python tools/step_gaps.py <rocprof output dir>"""
import csv
import glob
import os
import statistics
import sys

rows = []
for fn in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(fn) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0][-40:]))
rows.sort()
fin = [i for i, r in enumerate(rows) if "finish_step" in r[2]]
across, inside, step_wall, step_busy = [], [], [], []
for a, b in zip(fin, fin[1:]):
    if b - a < 50 or b - a > 400:            # a decode step of the captured graph: 99 .. 228 launches
        continue
    across.append((rows[a + 1][0] - rows[a][1]) / 1e3)
    g = [(rows[i + 1][0] - rows[i][1]) / 1e3 for i in range(a + 1, b)]
    inside.append(sum(g))
    step_wall.append((rows[b][1] - rows[a][1]) / 1e3)
    step_busy.append(sum(rows[i][1] - rows[i][0] for i in range(a + 1, b + 1)) / 1e3)
if not across:
    print("no decode steps found"); sys.exit(0)
print(f"{len(across)} decode steps: step wall (finish -> finish) median {statistics.median(step_wall):.1f} us, kernel durations {statistics.median(step_busy):.1f} us")
print(f"gap across the step boundary: median {statistics.median(across):.2f} us, mean {statistics.mean(across):.2f}, p90 {sorted(across)[int(0.9 * len(across))]:.2f}")
print(f"sum of the gaps inside a step: median {statistics.median(inside):.2f} us")
import collections
hist = collections.Counter(round(x) for x in across)
print("gap across the boundary, rounded to us -> steps:", sorted(hist.items()))
