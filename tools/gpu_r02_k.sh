#!/usr/bin/env bash
# round 2, call K: the rest of the GPU suite (ops, serving, minlen) + runtime launch-path knobs A/B on the decode step
set -u
OUT="gpurun_out/r02k"
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_minlen.py tests/test_gpu_ops.py tests/test_gpu_serving.py -m gpu -q 2>&1 | tail -25 > "$OUT/pytest_gpu_rest.log"
cat "$OUT/pytest_gpu_rest.log"
run() {   # label, env...
  local label="$1"; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 1 --warmup 1 --new-tokens 512 --ttft-requests 4 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', d['value'], 'tok/s', d['decode_us_per_step'], 'us/step ttft', d['ttft_p50_ms'])" \
    | tee -a "$OUT/runtime_knobs_ab.log"
}
run baseline A=1
run dev_kernarg_1 HIP_FORCE_DEV_KERNARG=1
run dev_kernarg_0 HIP_FORCE_DEV_KERNARG=0
run graph_capture_0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run graph_capture_1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run opt_flush_0 AMD_OPT_FLUSH=0
run sys_scope_signal_0 ROC_SYSTEM_SCOPE_SIGNAL=0
run fgs_kernarg_0 ROC_USE_FGS_KERNARG=0
run graph_batch_512 DEBUG_HIP_GRAPH_BATCH_SIZE=512
run baseline_again A=1
