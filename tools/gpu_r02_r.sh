#!/usr/bin/env bash
set -u
OUT="gpurun_out/r02r"
mkdir -p "$OUT"
export TMPDIR=/tmp
python tools/diag/cols_ln_diag.py > "$OUT/cols_ln_diag.log" 2>&1; cat "$OUT/cols_ln_diag.log" | cut -c1-330
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED" > "$OUT/ops_file.log"; cat "$OUT/ops_file.log"
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "decode_cols" 2>&1 | grep -E "passed|failed|FAILED" > "$OUT/ops_cols_only.log"; cat "$OUT/ops_cols_only.log"
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "preprocess or decode_cols" 2>&1 | grep -E "passed|failed|FAILED" > "$OUT/ops_pp_cols.log"; cat "$OUT/ops_pp_cols.log"
