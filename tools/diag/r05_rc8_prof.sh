for d in ${DELAYS:-0 400}; do
  bash tools/gpu_round_start.sh r05p8c_$d env:SV_RC_DELAY=$d "prof:8brc:bench.py --model 8b --new-tokens 128 --steps 1 --warmup 1 --no-cpu-baseline --ttft-requests 1" 2>&1 | grep -i "rowln\|row_update\|attn_decode" | head -4
done
