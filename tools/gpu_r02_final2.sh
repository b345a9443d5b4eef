#!/usr/bin/env bash
# round 2, final evidence 2: HBM bytes of the dominant kernel (separate FETCH_SIZE / WRITE_SIZE passes), MFMA-busy counters of the
# prefill kernels, and BASELINE configs 4 / 5 workloads on the final code
set -u
OUT="gpurun_out/r02final2"
mkdir -p "$OUT"
export TMPDIR=/tmp
bash tools/box_info.sh 2>&1 | grep -E "Unique ID" | tee "$OUT/box.txt"
pmc() {  # name, counters...
  local name="$1"; shift
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/pmc_$name" -- \
      python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --new-tokens 16 --no-cpu-baseline --ttft-requests 1 \
      > "$GRAFT_REPO_ROOT/$OUT/pmc_$name.bench.json" 2> "$GRAFT_REPO_ROOT/$OUT/pmc_$name.err" )
  python tools/pmc_summary.py "$OUT/pmc_$name" "$OUT/pmc_$name.json" 2>&1 | tail -1
  rm -rf "$OUT/pmc_$name"
}
pmc fetch_size FETCH_SIZE
pmc write_size WRITE_SIZE
pmc mfma_util SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
timeout 420 python bench.py --model 8b --new-tokens 256 --steps 2 --no-cpu-baseline > "$OUT/bench_8b.json" 2> "$OUT/bench_8b.err"
timeout 420 python bench.py --model 8b --weights fp8 --task text2svg --new-tokens 256 --steps 2 --no-cpu-baseline > "$OUT/bench_8b_fp8_text2svg.json" 2> "$OUT/bench_8b_fp8_text2svg.err"
timeout 420 python bench.py --model 8b --task text2svg --new-tokens 256 --steps 2 --no-cpu-baseline > "$OUT/bench_8b_bf16_text2svg.json" 2> "$OUT/bench_8b_bf16_text2svg.err"
for f in bench_8b bench_8b_fp8_text2svg bench_8b_bf16_text2svg; do python - "$OUT/$f.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], d["value"], "tok/s", d["decode_us_per_step"], "us/step ttft", d["ttft_p50_ms"], d["roofline"]["frac"])
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
done
