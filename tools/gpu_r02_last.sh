#!/usr/bin/env bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r02last
bash tools/box_info.sh 2>&1 | grep -E "Unique ID"
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_serving.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED" | tee gpurun_out/r02last/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/r02last/smoke.log
