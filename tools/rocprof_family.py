#!/usr/bin/env python3
"""profiles/rocprof_family.json: the rocprofv3 --kernel-trace figure of the decoder's weight-streaming GEMM family of the decode step (the `roofline.kernel`
of bench.py's line: gemm_skinny_kernel<*> + gemm_cols_resid_kernel + mlp_fused_kernel + rowln_cattn_kernel), from a by-(kernel, grid) summary
(tools/trace_by_grid.py output):
    python tools/rocprof_family.py profiles/rocprof_r06_by_grid.csv "<where measured>"
bench.py quotes it beside its own in-situ figure (the in-situ figure is a subtraction of chains; this one is the profiler's)."""
import csv
import json
import sys

FAMILY = ("gemm_skinny_kernel", "gemm_head_persist_kernel", "gemm_cols_resid_kernel", "mlp_fused_kernel", "rowln_cattn_kernel")
rows = [r for r in csv.DictReader(open(sys.argv[1])) if any(f in r["kernel"] for f in FAMILY)]
calls = sum(int(r["calls"]) for r in rows)
total = sum(float(r["total_us"]) for r in rows)
steps = max(int(r["calls"]) for r in rows if "mlp_fused_kernel" in r["kernel"] or "gemm_skinny" in r["kernel"] or "gemm_head_persist" in r["kernel"]) if rows else 0
out = {"family": list(FAMILY), "launches": calls, "total_us": round(total, 1), "avg_launch_us": round(total / max(calls, 1), 3),
       "by_kernel": {f'{r["kernel"]} grid {r["grid_x"]}x{r["grid_y"]}': {"calls": int(r["calls"]), "avg_us": float(r["avg_us"])} for r in rows},
       "source": sys.argv[1], "measured": sys.argv[2] if len(sys.argv) > 2 else ""}
json.dump(out, open("profiles/rocprof_family.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("launches", "total_us", "avg_launch_us")}))
