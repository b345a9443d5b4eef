#!/usr/bin/env python3
"""Group a rocprofv3 kernel_trace.csv by (kernel, grid, workgroup): calls / avg / min / max duration (us)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d, out = sys.argv[1], sys.argv[2]
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    agg = defaultdict(list)
    for fn in files:
        with open(fn) as f:
            for r in csv.DictReader(f):
                name = r.get("Kernel_Name", "")
                short = name.split("(")[0][-60:]
                grid = (r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
                wg = r.get("Workgroup_Size_X", r.get("Workgroup_Size", ""))
                dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                agg[(short, grid, wg)].append(dur)
    rows = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
    with open(out, "w") as f:
        f.write("kernel,grid_x,grid_y,grid_z,wg,calls,total_us,avg_us,min_us,max_us\n")
        for (name, grid, wg), v in rows[:60]:
            f.write('"%s",%s,%s,%s,%s,%d,%.1f,%.2f,%.2f,%.2f\n' % (name, grid[0], grid[1], grid[2], wg, len(v), sum(v),
                                                               sum(v) / len(v), min(v), max(v)))
    print("wrote", out)


if __name__ == "__main__":
    main()
