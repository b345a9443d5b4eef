#!/usr/bin/env bash
# round 2, call L: two-row-tile decode GEMMs (batch 64), serving / min-length fixes, 2-rank run of bench.py on one GPU
set -u
OUT="gpurun_out/r02l"
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_serving.py tests/test_gpu_minlen.py "tests/test_gpu_e2e.py::test_full_size_properties_batch32" "tests/test_gpu_e2e.py::test_more_than_one_row_tile_and_short_prompts" -m gpu -q 2>&1 | tail -25 > "$OUT/pytest_gpu_subset.log"
cat "$OUT/pytest_gpu_subset.log"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 1 --warmup 1 --new-tokens 256 --ttft-requests 2 > "$OUT/bench_2ranks_shared_gpu.json" 2> "$OUT/bench_2ranks.err"
tail -3 "$OUT/bench_2ranks.err"; cat "$OUT/bench_2ranks_shared_gpu.json"
timeout 900 python tools/bench_mt2.py fp8 bf16 > "$OUT/mt2_ab_8b_text2svg_b64.log" 2>&1
cat "$OUT/mt2_ab_8b_text2svg_b64.log"
