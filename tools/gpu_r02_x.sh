#!/usr/bin/env bash
# round 2, call X: the 8-wave LayerNorm-prologue GEMM in the context that broke the 16-wave one (x3), then the whole GPU suite
set -u
OUT="gpurun_out/r02x"
mkdir -p "$OUT"
export TMPDIR=/tmp
bash tools/box_info.sh 2>&1 | grep -E "Unique ID" | tee "$OUT/box.txt"
for rep in 1 2 3; do
  timeout 300 python -m pytest tests/test_gpu_minlen.py tools/diag/test_diag_cols2.py -m gpu -q -s 2>&1 | grep -E "engine #|passed|failed" | tr '\n' ' '; echo
done | tee "$OUT/cols_ln_8wave_first_launch.log"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error" | head -20 | tee "$OUT/pytest_gpu.log"
