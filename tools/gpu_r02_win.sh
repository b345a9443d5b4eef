#!/usr/bin/env bash
set -u
OUT="gpurun_out/r02win"
mkdir -p "$OUT"
export TMPDIR=/tmp
bash tools/box_info.sh 2>&1 | grep -E "Unique ID" | tee "$OUT/box.txt"
timeout 600 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_serving.py tests/test_gpu_ops.py -m gpu -q -s -k "sliding_window or attention or serving or 8b or padded or join" 2>&1 | grep -E "window\]|passed|failed|FAILED|Error" | tee "$OUT/pytest_window_serving.log"
