#!/usr/bin/env bash
# round 2, call J: full GPU suite on the LDS-epilogue + autotuned GEMM build, the GEMM A/B, bench with TTFT
set -u
OUT="gpurun_out/r02j"
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > "$OUT/pytest_gpu.log"
cat "$OUT/pytest_gpu.log"
SV_GEMM_AUTOTUNE_LOG=1 timeout 400 python tools/bench_gemm_epi.py > "$OUT/gemm_epilogue_ab.log" 2>&1
cat "$OUT/gemm_epilogue_ab.log"
SV_GEMM_AUTOTUNE_LOG=1 timeout 600 python bench.py --no-cpu-baseline --steps 2 --ttft-requests 20 > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -20 "$OUT/bench.err"
cat "$OUT/bench.json"
SV_GEMM_AUTOTUNE=0 timeout 600 python bench.py --no-cpu-baseline --steps 2 --ttft-requests 20 > "$OUT/bench_model_dispatch.json" 2>/dev/null
cat "$OUT/bench_model_dispatch.json"
