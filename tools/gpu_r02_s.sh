#!/usr/bin/env bash
# round 2, call S: prefill attention with K/V tiles loaded one tile ahead; whole GPU suite; bench
set -u
OUT="gpurun_out/r02s"
mkdir -p "$OUT"
export TMPDIR=/tmp
bash tools/box_info.sh > "$OUT/box.txt" 2>&1; cat "$OUT/box.txt"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|Error" | head -20 > "$OUT/pytest_gpu.log"
cat "$OUT/pytest_gpu.log"
timeout 600 python bench.py --no-cpu-baseline --steps 2 --ttft-requests 20 > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -3 "$OUT/bench.err"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r02s/bench.json").read().strip().splitlines()[-1])
print(d["value"], "tok/s", d["decode_us_per_step"], "us/step ttft", d["ttft_p50_ms"], d["roofline_prefill_gemm"]["us_per_layer"], d["decode_step_profile_ms"])
PY
