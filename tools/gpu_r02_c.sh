#!/usr/bin/env bash
# round 2, call C: parity closure -- re-minted fixtures (designed streams), 130-token 1B test, real-bf16 comparator, smoke
set -u
OUT="gpurun_out/r02c"
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x -s -k "goldens or semantics or streaming or 1b_shapes or real_bf16 or alternative" 2>&1 | grep -v Warning | tail -60 > "$OUT/pytest_parity.log"
cat "$OUT/pytest_parity.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
tail -5 "$OUT/smoke.log"
