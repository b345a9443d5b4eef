#!/usr/bin/env bash
set -u
OUT="gpurun_out/r02bar"
mkdir -p "$OUT"
bash tools/box_info.sh 2>&1 | grep -E "Unique ID" | tee "$OUT/box.txt"
timeout 120 ./tools/diag/grid_barrier_bench 2>&1 | tee "$OUT/grid_barrier.log"
