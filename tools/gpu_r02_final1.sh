#!/usr/bin/env bash
# round 2, final evidence 1: whole GPU suite, the driver-contract bench line, per-kernel time of the same command
set -u
OUT="gpurun_out/r02final1"
mkdir -p "$OUT"
export TMPDIR=/tmp
bash tools/box_info.sh > "$OUT/box.txt" 2>&1; grep -E "Unique ID" "$OUT/box.txt"
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -12 > "$OUT/pytest_gpu.log"; tail -4 "$OUT/pytest_gpu.log"
timeout 600 python bench.py > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"; cat "$OUT/bench_n1.json"
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/rocprof" -- \
    python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --steps 1 --warmup 1 --ttft-requests 2 \
    > "$GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.json" 2> "$GRAFT_REPO_ROOT/$OUT/rocprof.err" )
python tools/rocprof_summary.py "$OUT/rocprof" "$OUT/rocprof_kernel_stats.csv" > "$OUT/rocprof_summary.log" 2>&1 || true
find "$OUT/rocprof" -name '*kernel_trace.csv' -size +8M -delete 2>/dev/null
head -12 "$OUT/rocprof_kernel_stats.csv" | cut -c1-150
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee "$OUT/smoke.log"
