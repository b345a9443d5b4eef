#!/usr/bin/env bash
# round 2: kernel trace of a short bench run (256 new tokens), grouped by (kernel, grid)
set -u
OUT="gpurun_out/r02prof1"
mkdir -p "$OUT"
export TMPDIR=/tmp
bash tools/box_info.sh 2>&1 | grep -E "Unique ID" | tee "$OUT/box.txt"
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/rocprof" -- \
    python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --steps 1 --warmup 1 --new-tokens 256 --ttft-requests 4 \
    > "$GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.json" 2> "$GRAFT_REPO_ROOT/$OUT/rocprof.err" )
python tools/rocprof_summary.py "$OUT/rocprof" "$OUT/rocprof_kernel_stats.csv" > "$OUT/rocprof_summary.log" 2>&1 || true
python tools/trace_by_grid.py "$OUT/rocprof" "$OUT/rocprof_by_grid.csv" > "$OUT/by_grid.log" 2>&1 || true
find "$OUT/rocprof" -name '*kernel_trace.csv' -size +8M -delete 2>/dev/null
head -30 "$OUT/rocprof_kernel_stats.csv" | cut -c1-160
head -40 "$OUT/rocprof_by_grid.csv" | cut -c1-200
