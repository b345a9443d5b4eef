#!/usr/bin/env bash
# Copies what tools/diag/round_final.sh left under gpurun_out/<tag>/ into profiles/ under the round's names and regenerates the two derived
# files bench.py quotes (profiles/rocprof_family.json, profiles/hbm_traffic.json).   usage: bash tools/diag/file_final.sh <tag> "<session label>"
set -e
TAG="$1"; LABEL="${2:-round 6 final code}"; O=gpurun_out/$TAG; P=profiles
cp $O/bench_n1.json $P/bench_r06_n1.json
cp $O/bench_8b_im2svg.json $P/bench_r06_8b_im2svg_n1_256tok.json
cp $O/bench_8b_fp8_text2svg.json $P/bench_r06_8b_fp8_text2svg_n1_256tok.json
cp $O/bench_beam2_n1.json $P/bench_r06_beam2_n1.json
cp $O/bench_n1_256tok.json $P/bench_r06_n1_256tok.json
cp $O/bench_n1_4096tok.json $P/bench_r06_n1_4096tok.json
cp $O/bench_under_rocprof.json $P/bench_r06_n1_under_rocprof.json
grep '^{' $O/bench_2ranks_one_gpu_gloo.json | tail -1 > $P/bench_r06_2ranks_one_gpu_gloo.json      # (gloo prints its rendezvous lines on stdout in front of the JSON line)
cp $O/rocprof_kernel_stats.csv $P/rocprof_r06_kernel_stats.csv
cp $O/rocprof_by_grid.csv $P/rocprof_r06_by_grid.csv
cp $O/pmc_FETCH_SIZE.json $P/pmc_r06_fetch_size.json
cp $O/pmc_WRITE_SIZE.json $P/pmc_r06_write_size.json
cp $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES.json $P/pmc_r06_mfma_busy.json
cp $O/ctx_sweep.log $P/ctx_sweep_r06.log
cp $O/prefill_ceiling.log $P/prefill_ceiling_r06.log
cp $O/gemm_autotune_choices.log $P/gemm_autotune_choices_r06.log
cp $O/pytest_gpu.log $P/pytest_gpu_r06_final.log
cp $O/smoke.log $P/smoke_r06.log
cp $O/box.txt $P/box_r06_final.txt
cp $O/step_gaps.log $P/step_gaps_r06_final_tree.log
cp $O/ttft_gaps.log $P/ttft_gaps_r06_final_tree.log
python tools/rocprof_family.py $P/rocprof_r06_by_grid.csv "$LABEL, rocprofv3 --kernel-trace of python bench.py --steps 1 --warmup 1 (gpurun_out/$TAG)"
ALG=$(python -c "import json; print(json.load(open('$P/bench_r06_n1.json'))['roofline']['algorithmic_bytes_per_launch'])")
python tools/hbm_traffic.py $P/pmc_r06_fetch_size.json $P/pmc_r06_write_size.json $ALG "$LABEL (tools/diag/round_final.sh pmc, gpurun_out/$TAG; profiles/pmc_r06_fetch_size.json, pmc_r06_write_size.json)" > $P/hbm_traffic.json
python - <<'PY'
import json
j = json.load(open('profiles/hbm_traffic.json'))
j.setdefault('launches_per_step', 73)      # the family of an engine that owns its GPU: 24 x (rowln_cattn + cols + mlp_fused) + lm_head (bench.py quotes the figure for that family only)
json.dump(j, open('profiles/hbm_traffic.json', 'w'), indent=1)
PY
