#!/usr/bin/env bash
# round 2: four register chunks in flight per wave in the decode GEMMs (SV_SKINNY_DEPTH=4) against two, in situ
set -u
OUT="gpurun_out/r02depth"
mkdir -p "$OUT"
export TMPDIR=/tmp
bash tools/box_info.sh 2>&1 | grep -E "Unique ID" | tee "$OUT/box.txt"
run() {
  local label="$1"; shift
  env "$@" timeout 200 python bench.py --no-cpu-baseline --steps 1 --warmup 1 --new-tokens 384 --ttft-requests 2 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', d['value'], 'tok/s', d['decode_us_per_step'], 'us/step', d['decode_step_profile_ms'])" \
    | tee -a "$OUT/skinny_depth_ab.log"
}
run depth2 A=1
run depth4 SV_SKINNY_DEPTH=4
run depth2_again A=1
run depth4_again SV_SKINNY_DEPTH=4
SV_SKINNY_DEPTH=4 timeout 200 python -m pytest tests/test_gpu_e2e.py -m gpu -q -k "1b_shapes or full_size_properties_batch32" 2>&1 | grep -E "passed|failed" | tee -a "$OUT/skinny_depth_ab.log"
