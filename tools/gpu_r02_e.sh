#!/usr/bin/env bash
# round 2, call E: serving tests again; idle-block weight prefetch A/B
set -u
OUT="gpurun_out/r02e"
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_serving.py -m gpu -q -x 2>&1 | grep -v Warning | tail -30 > "$OUT/pytest_serving.log"
tail -12 "$OUT/pytest_serving.log"
timeout 300 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -k "deterministic or full_size or more_than_one" 2>&1 | tail -5
for cfg in "0 64" "1 32" "1 64" "1 128" "0 64" "1 64"; do
  set -- $cfg
  SV_PREFETCH=$1 SV_PREFETCH_KB=$2 timeout 200 python bench.py --no-cpu-baseline --steps 2 --ttft-requests 2 > "$OUT/bench_pf$1_kb$2.json" 2> "$OUT/bench_pf$1_kb$2.err"
  python - "$OUT/bench_pf$1_kb$2.json" "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("prefetch", sys.argv[2], "kb", sys.argv[3], ":", d["value"], "tok/s", d["decode_us_per_step"], "us/step", d["roofline"]["avg_launch_us"], d["decode_step_profile_ms"])
except Exception as e:
    print("failed", sys.argv[1:], e)
PY
done
