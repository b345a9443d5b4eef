#!/usr/bin/env bash
set -u
OUT="gpurun_out/r02u"
mkdir -p "$OUT"
export TMPDIR=/tmp
bash tools/box_info.sh 2>&1 | grep -E "Unique ID"
python tools/diag/cols_ln_diag.py 2>&1 | grep -v amdgpu.ids | cut -c1-420 | tee "$OUT/cols_ln_diag.log"
