#!/usr/bin/env bash
set -u
OUT="gpurun_out/r02ab"
mkdir -p "$OUT"
export TMPDIR=/tmp
bash tools/box_info.sh 2>&1 | grep -E "Unique ID" | tee "$OUT/box.txt"
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "linear" 2>&1 | tail -2
timeout 300 python tools/bench_gemm_order.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/gemm_tile_order_ab.log"
SV_GEMM_AUTOTUNE_LOG=1 timeout 400 python bench.py --no-cpu-baseline --steps 1 --warmup 1 --new-tokens 256 --ttft-requests 20 2> "$OUT/bench.err" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ttft', d['ttft_p50_ms'], 'prefill layer us', d['roofline_prefill_gemm']['us_per_layer'], d['roofline_prefill_gemm']['gemms'])" | tee "$OUT/bench_ttft.log"
grep autotune "$OUT/bench.err" | tee "$OUT/autotune.log"
