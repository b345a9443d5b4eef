#!/usr/bin/env bash
set -u
OUT="gpurun_out/r02g"
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "linear or layernorm" 2>&1 | tail -8 > "$OUT/pytest_gemm_ln.log"
tail -4 "$OUT/pytest_gemm_ln.log"
timeout 600 python -m pytest tests/test_gpu_serving.py tests/test_gpu_minlen.py -m gpu -q -s 2>&1 | grep -v Warning | tail -12 > "$OUT/pytest_serving_minlen.log"
tail -8 "$OUT/pytest_serving_minlen.log"
timeout 400 python tools/bench_gemm_epi.py > "$OUT/gemm_epilogue_ab.log" 2>&1
cat "$OUT/gemm_epilogue_ab.log"
timeout 300 python bench.py --no-cpu-baseline --steps 2 --ttft-requests 20 > "$OUT/bench.json" 2> "$OUT/bench.err"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r02g/bench.json").read().strip().splitlines()[-1])
print(d["value"], "tok/s", d["decode_us_per_step"], "us/step ttft", d["ttft_p50_ms"], d["roofline_prefill_gemm"])
PY
