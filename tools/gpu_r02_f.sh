#!/usr/bin/env bash
set -u
OUT="gpurun_out/r02f"
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_serving.py -m gpu -q -x 2>&1 | grep -v Warning | tail -30 > "$OUT/pytest_serving.log"
tail -12 "$OUT/pytest_serving.log"
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -s -k "8b_dims" 2>&1 | grep -v Warning | tail -30 > "$OUT/pytest_8bdims.log"
tail -12 "$OUT/pytest_8bdims.log"
