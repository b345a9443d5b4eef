#!/usr/bin/env bash
# round 2, call N: whole GPU suite on the hardware-rounding / reciprocal-activation build, fixed-cost sweep again, bench
set -u
OUT="gpurun_out/r02n"
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -15 > "$OUT/pytest_gpu.log"
cat "$OUT/pytest_gpu.log"
timeout 500 python tools/bench_gemm_fixed_cost.py > "$OUT/gemm_fixed_cost_sweep.log" 2>&1
cat "$OUT/gemm_fixed_cost_sweep.log"
SV_GEMM_AUTOTUNE_LOG=1 timeout 600 python bench.py --no-cpu-baseline --steps 2 --ttft-requests 20 > "$OUT/bench.json" 2> "$OUT/bench.err"
grep autotune "$OUT/bench.err"
cat "$OUT/bench.json"
