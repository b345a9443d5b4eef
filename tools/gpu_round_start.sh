#!/usr/bin/env bash
# ONE parameterised script for the GPU box (replaces round 2's 41 one-shot scripts).  Usage, from the repo root on the box:
#   bash tools/gpu_round_start.sh <tag> [steps...]      steps (default "pytest bench"), run in the order given:
#     pytest      the whole GPU suite through the C ABI            -> gpurun_out/<tag>/pytest_gpu.log
#     pytestall   the same without -x (every failure of a first run of new tests)
#     pytest:<k>  only tests matching -k <k>
#     bench       the headline line (BASELINE config 2)            -> bench_n1.json
#     bench8b     configs 4 and 5 per-GPU workloads (256 tokens)   -> bench_8b_*.json
#     rocprof     rocprofv3 --kernel-trace --stats of the headline -> rocprof_kernel_stats.csv, rocprof_by_grid.csv
#     pmc         FETCH_SIZE / WRITE_SIZE / MFMA-busy passes        -> pmc_*.json  (kernel-trace only: never with sys-trace)
#     smoke       __graft_entry__.smoke()
#     ctx         tools/ctx_sweep.py                                -> ctx_sweep.log
#     skinny      tools/bench_skinny.py                             -> bench_skinny.log
#     memmix      tools/diag/mem_mix.hip: weight stream + shared-operand re-reads without MFMA (what bounds the decode GEMMs) -> mem_mix.log
#     xcd         tools/diag/xcd_map.hip: block -> XCD placement inside a replayed graph  -> xcd_map.log
#     ab:<args>   tools/ab_exp.py <args> (in-process A/B of SV_EXP masks)           -> ab_exp.log
#     run:<name>:<script args>  python <script args> -> <name>.log
#     rebuild:<K=V>  rebuild the library on the box with K=V in the environment (build-flag A/B, e.g. SV_NO_KERNARG_PRELOAD=1)
#     env:<K=V>   export K=V for the steps that follow (experiment switches)
# Everything lands under gpurun_out/<tag>/ ; copy what is to be judged into profiles/.
set -u
TAG="${1:-rXX}"; shift || true
STEPS=("$@"); [ ${#STEPS[@]} -eq 0 ] && STEPS=(pytest bench)
OUT="gpurun_out/${TAG}"
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT="${GRAFT_REPO_ROOT:-$PWD}"
bash tools/box_info.sh > "$OUT/box.txt" 2>&1        # results have differed between the pool's GPUs: keep the identity
for s in "${STEPS[@]}"; do
  case "$s" in
    env:*) export "${s#env:}"; echo "[env] ${s#env:}" ;;
    # -s -rA: the parity tests PRINT what they measured (max|err|, scale, checked / near-tie counts, comparator ratios); the numbers are
    # the evidence, so the full log is kept and the "[tag] ..." lines are cut out next to it
    pytest) timeout 1500 python -m pytest tests -m gpu -q -x -s -rA 2>&1 | grep -v amdgpu.ids > "$OUT/pytest_gpu_full.log"
            grep -oE "\[[A-Za-z0-9][^]]*\] .*|^.*(passed|failed).*$|^(FAILED|ERROR).*$" "$OUT/pytest_gpu_full.log" > "$OUT/pytest_gpu.log"; tail -4 "$OUT/pytest_gpu.log" ;;
    pytestall) timeout 2400 python -m pytest tests -m gpu -q -s -rA 2>&1 | grep -v amdgpu.ids > "$OUT/pytest_gpu_full.log"       # no -x: every failure of a first run
            grep -oE "\[[A-Za-z0-9][^]]*\] .*|^.*(passed|failed).*$|^(FAILED|ERROR).*$" "$OUT/pytest_gpu_full.log" > "$OUT/pytest_gpu.log"; grep -E "^(FAILED|ERROR)|passed|failed" "$OUT/pytest_gpu.log" | tail -30 ;;
    pytest:*) timeout 1500 python -m pytest tests -m gpu -q -x -s -rA -k "${s#pytest:}" 2>&1 | grep -v amdgpu.ids > "$OUT/pytest_gpu_k_full.log"
            grep -oE "\[[A-Za-z0-9][^]]*\] .*|^.*(passed|failed|Error|assert).*$|^(FAILED|ERROR).*$" "$OUT/pytest_gpu_k_full.log" > "$OUT/pytest_gpu_k.log"; tail -25 "$OUT/pytest_gpu_k.log" ;;
    bench) timeout 400 python bench.py > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"; cat "$OUT/bench_n1.json" ;;
    bench8b)
      timeout 300 python bench.py --model 8b --new-tokens 256 --steps 2 --no-cpu-baseline > "$OUT/bench_8b_im2svg.json" 2> "$OUT/bench_8b_im2svg.err"
      timeout 300 python bench.py --model 8b --weights fp8 --task text2svg --new-tokens 256 --steps 2 --no-cpu-baseline \
          > "$OUT/bench_8b_fp8_text2svg.json" 2> "$OUT/bench_8b_fp8_text2svg.err"
      cat "$OUT/bench_8b_im2svg.json" "$OUT/bench_8b_fp8_text2svg.json" ;;
    rocprof)
      ( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/rocprof" -- \
          python "$ROOT/bench.py" --no-cpu-baseline --steps 1 --warmup 1 --ttft-requests 2 \
          > "$ROOT/$OUT/bench_under_rocprof.json" 2> "$ROOT/$OUT/rocprof.err" )
      python tools/rocprof_summary.py "$OUT/rocprof" "$OUT/rocprof_kernel_stats.csv" > "$OUT/rocprof_summary.log" 2>&1 || true
      python tools/trace_by_grid.py "$OUT/rocprof" "$OUT/rocprof_by_grid.csv" > "$OUT/by_grid.log" 2>&1 || true
      find "$OUT/rocprof" -name '*kernel_trace.csv' -size +8M -delete 2>/dev/null   # keep the pull under 64 MiB
      head -14 "$OUT/rocprof_kernel_stats.csv" | cut -c1-160 ;;
    pmc)
      for ctr in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
        tagc="${ctr%% *}"
        ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$ROOT/$OUT/pmc_$tagc" -- \
            python "$ROOT/bench.py" --no-cpu-baseline --steps 1 --warmup 0 --new-tokens 64 --ttft-requests 1 \
            > /dev/null 2> "$ROOT/$OUT/pmc_$tagc.err" )
        python tools/pmc_summary.py "$OUT/pmc_$tagc" "$OUT/pmc_$tagc.json" > /dev/null 2>&1 || true
        find "$OUT/pmc_$tagc" -name '*.csv' -size +8M -delete 2>/dev/null
      done ;;
    smoke) python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee "$OUT/smoke.log" ;;
    ctx) timeout 400 python tools/ctx_sweep.py 2>&1 | tee "$OUT/ctx_sweep.log" ;;
    skinny) timeout 200 python tools/bench_skinny.py 2>&1 | tee "$OUT/bench_skinny.log" ;;
    rebuild:*) env ${s#rebuild:} python star-vector_amd/build.py --force 2>&1 | tail -1 ;;
    prof:*)   # rocprofv3 kernel trace of an arbitrary python command line, grouped by (kernel, grid): prof:<name>:<script and args>
      rest="${s#prof:}"; name="${rest%%:*}"; cmd="${rest#*:}"
      ( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/rocprof_$name" -- \
          python $ROOT/$cmd > "$ROOT/$OUT/prof_$name.out" 2> "$ROOT/$OUT/prof_$name.err" )
      python tools/trace_by_grid.py "$OUT/rocprof_$name" "$OUT/rocprof_${name}_by_grid.csv" > /dev/null 2>&1 || true
      find "$OUT/rocprof_$name" -name '*kernel_trace.csv' -size +8M -delete 2>/dev/null
      head -14 "$OUT/rocprof_${name}_by_grid.csv" | cut -c1-150 ;;
    run:*)    # an arbitrary python script of the repo: run:<name>:<script and args>   -> <name>.log
      rest="${s#run:}"; name="${rest%%:*}"; cmd="${rest#*:}"
      timeout 600 python $cmd 2>&1 | grep -v amdgpu.ids | tee "$OUT/$name.log" | tail -40 ;;
    memmix) hipcc --offload-arch=gfx950 -O3 -o /tmp/mem_mix tools/diag/mem_mix.hip 2>/dev/null && timeout 120 /tmp/mem_mix 2>&1 | tee "$OUT/mem_mix.log" | tail -30 ;;
    xcd) hipcc --offload-arch=gfx950 -O2 -o /tmp/xcd_map tools/diag/xcd_map.hip 2>/dev/null && /tmp/xcd_map 2>&1 | tee "$OUT/xcd_map.log" | tail -8 ;;
    ab:*) timeout 600 python tools/ab_exp.py ${s#ab:} 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab_exp.log" ;;
    *) echo "unknown step $s" ;;
  esac
done
