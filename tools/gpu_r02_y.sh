#!/usr/bin/env bash
set -u
OUT="gpurun_out/r02y"
mkdir -p "$OUT"
export TMPDIR=/tmp
bash tools/box_info.sh 2>&1 | grep -E "Unique ID" | tee "$OUT/box.txt"
for dbg in 0 0 2 1 3 0; do
  SV_COLS_DBG=$dbg timeout 300 python -m pytest tests/test_gpu_minlen.py tools/diag/test_diag_cols2.py -m gpu -q -s 2>&1 | grep -E "engine #|passed|failed" | tr '\n' ' ' | sed "s/^/dbg $dbg: /"; echo
done | tee "$OUT/cols_ln_dbg.log"
