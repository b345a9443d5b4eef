#!/usr/bin/env bash
# round 2: kernel trace of BASELINE config 5's per-GPU workload (8B text2svg, batch 64, fp8 weights), grouped by (kernel, grid)
set -u
OUT="gpurun_out/r02prof8b"
mkdir -p "$OUT"
export TMPDIR=/tmp
bash tools/box_info.sh 2>&1 | grep -E "Unique ID" | tee "$OUT/box.txt"
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/rocprof" -- \
    python "$GRAFT_REPO_ROOT/bench.py" --model 8b --weights fp8 --task text2svg --no-cpu-baseline --steps 1 --warmup 0 --new-tokens 64 --ttft-requests 1 \
    > "$GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.json" 2> "$GRAFT_REPO_ROOT/$OUT/rocprof.err" )
python tools/trace_by_grid.py "$OUT/rocprof" "$OUT/rocprof_by_grid.csv" > "$OUT/by_grid.log" 2>&1 || true
find "$OUT/rocprof" -name '*kernel_trace.csv' -size +8M -delete 2>/dev/null
head -24 "$OUT/rocprof_by_grid.csv" | cut -c1-200
