#!/usr/bin/env bash
# One gpurun call that re-establishes the round's baseline on a fresh MI355X box (about 6 GPU-minutes):
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_round_start.sh r02'
# Writes everything under gpurun_out/<tag>/ ; copy what is to be judged into profiles/.
set -u
TAG="${1:-rXX}"
OUT="gpurun_out/${TAG}"
mkdir -p "$OUT"
export TMPDIR=/tmp
bash tools/box_info.sh > "$OUT/box.txt" 2>&1        # results have differed between the pool's GPUs: keep the identity

# 1. parity: the whole GPU suite through the C ABI
timeout 400 python -m pytest tests -m gpu -q 2>&1 | tail -40 > "$OUT/pytest_gpu.log"

# 2. the headline line (BASELINE config 2) and the secondary ones
timeout 300 python bench.py > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
timeout 300 python bench.py --model 8b --new-tokens 256 --steps 2 --no-cpu-baseline > "$OUT/bench_8b.json" 2> "$OUT/bench_8b.err"
# BASELINE config 5's workload (first measurement: it went in after round 1's GPU minutes were spent)
timeout 300 python bench.py --model 8b --weights fp8 --task text2svg --new-tokens 256 --steps 2 --no-cpu-baseline \
    > "$OUT/bench_8b_fp8_text2svg.json" 2> "$OUT/bench_8b_fp8_text2svg.err"

# 3. per-kernel time of the headline command (kernel trace only: never together with --pmc)
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/rocprof" -- \
    python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --steps 1 --warmup 1 --ttft-requests 2 \
    > "$GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.json" 2> "$GRAFT_REPO_ROOT/$OUT/rocprof.err" )
python tools/rocprof_summary.py "$OUT/rocprof" "$OUT/rocprof_kernel_stats.csv" > "$OUT/rocprof_summary.log" 2>&1 || true
find "$OUT/rocprof" -name '*kernel_trace.csv' -size +8M -delete 2>/dev/null   # keep the pull under 64 MiB
tail -3 "$OUT/pytest_gpu.log"; cat "$OUT/bench_n1.json"
