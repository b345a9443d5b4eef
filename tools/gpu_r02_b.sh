#!/usr/bin/env bash
# round 2, call B: the full-K decode pipeline -- op tests, the whole GPU suite on it, A/B against the slab pipeline
set -u
OUT="gpurun_out/r02b"
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "decode_cols or decode_skinny" 2>&1 | tail -25 > "$OUT/pytest_new_ops.log"
tail -5 "$OUT/pytest_new_ops.log"
timeout 200 python tools/bench_decode_gemm.py 32 16 > "$OUT/bench_decode_gemm.log" 2>&1
cat "$OUT/bench_decode_gemm.log"
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40 > "$OUT/pytest_gpu_cols.log"
tail -15 "$OUT/pytest_gpu_cols.log"
timeout 300 python bench.py --no-cpu-baseline --steps 2 --ttft-requests 4 > "$OUT/bench_cols.json" 2> "$OUT/bench_cols.err"
SV_DECODE_PIPE=slabs timeout 300 python bench.py --no-cpu-baseline --steps 2 --ttft-requests 4 > "$OUT/bench_slabs.json" 2> "$OUT/bench_slabs.err"
python - <<'PY'
import json
for n in ("cols", "slabs"):
    try:
        d = json.loads(open(f"gpurun_out/r02b/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], "tok/s", d["decode_us_per_step"], "us/step", d["roofline"], d["decode_step_profile_ms"])
    except Exception as e:
        print(n, "failed", e)
PY
tail -3 "$OUT"/bench_cols.err "$OUT"/bench_slabs.err
