#!/usr/bin/env bash
set -u
OUT="gpurun_out/r02w"
mkdir -p "$OUT"
export TMPDIR=/tmp
bash tools/box_info.sh 2>&1 | grep -E "Unique ID"
echo "--- after test_gpu_minlen"; timeout 300 python -m pytest tests/test_gpu_minlen.py tools/diag/test_diag_cols2.py -m gpu -q -s 2>&1 | grep -E "diag2.|passed|failed" | tee "$OUT/after_minlen.log"
