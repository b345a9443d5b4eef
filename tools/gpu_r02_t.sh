#!/usr/bin/env bash
# round 2, call T: which earlier test file makes the in-block LayerNorm GEMM test fail later in the same process?
set -u
OUT="gpurun_out/r02t"
mkdir -p "$OUT"
export TMPDIR=/tmp
bash tools/box_info.sh 2>&1 | grep -E "Unique ID|sclk" 
T="tests/test_gpu_ops.py::test_decode_cols_layernorm_prologue"
for f in test_gpu_beam test_gpu_fp8 test_gpu_minlen test_gpu_e2e; do
  timeout 600 python -m pytest tests/$f.py "$T" -m gpu -q 2>&1 | grep -E "passed|failed|FAILED" | tr '\n' ' ' | sed "s/^/$f + cols-LN: /"; echo
done 2>&1 | tee "$OUT/bisect_files.log"
