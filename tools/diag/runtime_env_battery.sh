#!/usr/bin/env bash
# HIP runtime (ROCclr) knobs against the default on the headline bench, one process per arm, the default interleaved:  bash tools/diag/runtime_env_battery.sh <out.log>
OUT="${1:-/dev/stdout}"
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --ttft-requests 5 2>/dev/null | python3 -c '
import json,sys
for ln in sys.stdin:
    if ln.startswith("{\"metric\""):
        d=json.loads(ln); print("   decode %.1f us/step | ttft_p50 %.2f ms | value %.0f" % (d["decode_us_per_step"], d["ttft_p50_ms"], d["value"]))
'; }
{
for arm in "SV_NOP=1" "AMD_OPT_FLUSH=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "SV_NOP=1" "ROC_ACTIVE_WAIT_TIMEOUT=100000" "ROC_SYSTEM_SCOPE_SIGNAL=0" "ROC_USE_FGS_KERNARG=0" \
           "SV_NOP=1" "DEBUG_HIP_GRAPH_BATCH_SIZE=1" "DEBUG_HIP_GRAPH_BATCH_SIZE=1024" "DEBUG_HIP_FORCE_GRAPH_QUEUES=1" "DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0" "GPU_MAX_HW_QUEUES=1" "ROC_SKIP_KERNEL_ARG_COPY=1" "SV_NOP=1"; do
  echo "== $arm"; run "$arm"
done
} 2>&1 | tee "$OUT"
