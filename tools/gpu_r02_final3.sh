#!/usr/bin/env bash
# round 2, final evidence 3: decode step vs context length, the other decode lengths of SURVEY 8d
set -u
OUT="gpurun_out/r02final3"
mkdir -p "$OUT"
export TMPDIR=/tmp
bash tools/box_info.sh 2>&1 | grep -E "Unique ID" | tee "$OUT/box.txt"
timeout 400 python tools/ctx_sweep.py 2>/dev/null | tee "$OUT/ctx_sweep.log"
timeout 300 python bench.py --no-cpu-baseline --new-tokens 256 --steps 3 > "$OUT/bench_256tok.json" 2>/dev/null
timeout 400 python bench.py --no-cpu-baseline --new-tokens 4096 --steps 1 --ttft-requests 4 > "$OUT/bench_4096tok.json" 2>/dev/null
for f in bench_256tok bench_4096tok; do python - "$OUT/$f.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], d["value"], "tok/s", d["decode_us_per_step"], "us/step ttft", d["ttft_p50_ms"], d["decode_step_profile_ms"])
PY
done
