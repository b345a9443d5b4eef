#!/usr/bin/env bash
# round 2, call D: continuous batching, batched pre-processing, graph cache; then the whole suite
set -u
OUT="gpurun_out/r02d"
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_serving.py -m gpu -q -x 2>&1 | grep -v Warning | tail -40 > "$OUT/pytest_serving.log"
tail -25 "$OUT/pytest_serving.log"
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "preprocess" 2>&1 | tail -15 > "$OUT/pytest_preprocess.log"
tail -8 "$OUT/pytest_preprocess.log"
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -30 > "$OUT/pytest_gpu_all.log"
tail -12 "$OUT/pytest_gpu_all.log"
timeout 300 python bench.py --no-cpu-baseline --steps 3 --ttft-requests 6 > "$OUT/bench.json" 2> "$OUT/bench.err"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r02d/bench.json").read().strip().splitlines()[-1])
print(d["value"], "tok/s", d["decode_us_per_step"], "us/step ttft", d["ttft_p50_ms"], d["roofline"]["frac"], d["decode_step_profile_ms"])
PY
tail -3 "$OUT/bench.err"
