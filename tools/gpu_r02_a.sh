#!/usr/bin/env bash
# round 2, call A: BASELINE configs 4 and 5 on hardware (never measured in round 1), nothing else
set -u
OUT="gpurun_out/r02a"
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 420 python bench.py --model 8b --new-tokens 256 --steps 2 --no-cpu-baseline > "$OUT/bench_8b.json" 2> "$OUT/bench_8b.err"
timeout 420 python bench.py --model 8b --weights fp8 --task text2svg --new-tokens 256 --steps 2 --no-cpu-baseline \
    > "$OUT/bench_8b_fp8_text2svg.json" 2> "$OUT/bench_8b_fp8_text2svg.err"
timeout 300 python bench.py --model 8b --task text2svg --new-tokens 256 --steps 2 --no-cpu-baseline \
    > "$OUT/bench_8b_bf16_text2svg.json" 2> "$OUT/bench_8b_bf16_text2svg.err"
cat "$OUT/bench_8b.json" "$OUT/bench_8b_fp8_text2svg.json" "$OUT/bench_8b_bf16_text2svg.json"
tail -5 "$OUT"/*.err
