#!/usr/bin/env bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r02last3
timeout 60 python -m pytest tests/test_gpu_e2e.py -m gpu -q -s -k "sliding_window" 2>&1 | grep -E "window\]|passed|failed" | tee gpurun_out/r02last3/pytest.log
