#!/usr/bin/env bash
set -u
OUT="gpurun_out/r02v"
mkdir -p "$OUT"
export TMPDIR=/tmp
bash tools/box_info.sh 2>&1 | grep -E "Unique ID"
echo "--- alone"; timeout 300 python -m pytest tools/diag/test_diag_cols.py -m gpu -q -s 2>&1 | grep -E "diag|passed|failed" | tee "$OUT/alone.log"
echo "--- after test_gpu_minlen"; timeout 300 python -m pytest tests/test_gpu_minlen.py tools/diag/test_diag_cols.py -m gpu -q -s 2>&1 | grep -E "diag\]|passed|failed" | tee "$OUT/after_minlen.log"
