#!/usr/bin/env bash
# round 2, call M: two-row-tile GEMMs with the one-tile kernels' wave split (bitwise), their 8B A/B again, GEMM fixed-cost sweep
set -u
OUT="gpurun_out/r02m"
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py "tests/test_gpu_e2e.py::test_full_size_properties_batch32" "tests/test_gpu_e2e.py::test_more_than_one_row_tile_and_short_prompts" "tests/test_gpu_e2e.py::test_generate_is_deterministic_graph_equals_eager_and_batch_invariant" -m gpu -q 2>&1 | tail -8 > "$OUT/pytest_gpu_subset.log"
cat "$OUT/pytest_gpu_subset.log"
timeout 500 python tools/bench_gemm_fixed_cost.py > "$OUT/gemm_fixed_cost_sweep.log" 2>&1
cat "$OUT/gemm_fixed_cost_sweep.log"
timeout 900 python tools/bench_mt2.py fp8 bf16 > "$OUT/mt2_ab_8b_text2svg_b64.log" 2>&1
cat "$OUT/mt2_ab_8b_text2svg_b64.log"
