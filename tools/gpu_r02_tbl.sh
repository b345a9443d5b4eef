#!/usr/bin/env bash
set -u
OUT="gpurun_out/r02tbl"
mkdir -p "$OUT"
export TMPDIR=/tmp
bash tools/box_info.sh 2>&1 | grep -E "Unique ID" | tee "$OUT/box.txt"
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_beam.py tests/test_gpu_serving.py tests/test_gpu_minlen.py tests/test_gpu_fp8.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error" | tee "$OUT/pytest.log"
timeout 400 python bench.py --no-cpu-baseline --steps 2 --ttft-requests 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'tok/s', d['decode_us_per_step'], 'us/step ttft', d['ttft_p50_ms'], d['decode_step_profile_ms'])" | tee "$OUT/bench.log"
