#!/usr/bin/env bash
# round 2, call O: whole GPU suite after replacing the inline-asm bf16 conversion (hazard) by the compiler-selected one; bench
set -u
OUT="gpurun_out/r02o"
mkdir -p "$OUT"
export TMPDIR=/tmp
export PYTHONFAULTHANDLER=1
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -40 > "$OUT/pytest_gpu.log"
cat "$OUT/pytest_gpu.log"
SV_GEMM_AUTOTUNE_LOG=1 timeout 600 python bench.py --no-cpu-baseline --steps 2 --ttft-requests 20 > "$OUT/bench.json" 2> "$OUT/bench.err"
grep -c autotune "$OUT/bench.err"; grep -v autotune "$OUT/bench.err" | tail -5
cat "$OUT/bench.json"
