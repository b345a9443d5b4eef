#!/usr/bin/env bash
set -u
OUT="gpurun_out/r02nt2"
mkdir -p "$OUT"
export TMPDIR=/tmp
bash tools/box_info.sh 2>&1 | grep -E "Unique ID" | tee "$OUT/box.txt"
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "skinny" 2>&1 | tail -3
timeout 300 python tools/bench_skinny_nt2.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/skinny_nt2_ab.log"
