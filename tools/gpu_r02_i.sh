#!/usr/bin/env bash
set -u
OUT="gpurun_out/r02i"
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "test_linear" 2>&1 | tail -3
timeout 400 python tools/bench_gemm_epi.py > "$OUT/gemm_epilogue_ab.log" 2>&1
cat "$OUT/gemm_epilogue_ab.log"
