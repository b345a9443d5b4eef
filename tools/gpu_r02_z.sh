#!/usr/bin/env bash
set -u
OUT="gpurun_out/r02z"
mkdir -p "$OUT"
export TMPDIR=/tmp
bash tools/box_info.sh 2>&1 | grep -E "Unique ID" | tee "$OUT/box.txt"
run() { env "$@" timeout 300 python -m pytest tests/test_gpu_minlen.py tools/diag/test_diag_cols3.py -m gpu -q -s 2>&1 | grep -E "diag3" | sed "s/^/[$*] /"; }
{
run DIAG_MODE=base
run DIAG_MODE=base
run DIAG_MODE=keep
run DIAG_MODE=dummy
run DIAG_MODE=settle
run DIAG_MODE=base HIP_LAUNCH_BLOCKING=1
run DIAG_MODE=base AMD_SERIALIZE_KERNEL=3 AMD_SERIALIZE_COPY=3
run DIAG_MODE=base SV_COLS_WAVES=16
run DIAG_MODE=base
} | tee "$OUT/cols_ln_modes.log"
