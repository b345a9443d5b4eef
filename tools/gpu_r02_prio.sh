#!/usr/bin/env bash
set -u
OUT="gpurun_out/r02prio"
mkdir -p "$OUT"
export TMPDIR=/tmp
bash tools/box_info.sh 2>&1 | grep -E "Unique ID" | tee "$OUT/box.txt"
run() {
  local label="$1"; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 1 --warmup 1 --new-tokens 512 --ttft-requests 4 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', d['value'], 'tok/s', d['decode_us_per_step'], 'us/step ttft', d['ttft_p50_ms'])" \
    | tee -a "$OUT/stream_priority_ab.log"
}
run default A=1
run priority_high SV_STREAM_PRIORITY=high
run priority_low SV_STREAM_PRIORITY=low
run default_again A=1
run priority_high_again SV_STREAM_PRIORITY=high
