#!/usr/bin/env bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r02last2
bash tools/box_info.sh 2>&1 | grep -E "Unique ID"
timeout 200 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED" | tee gpurun_out/r02last2/pytest.log
