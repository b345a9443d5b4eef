#!/usr/bin/env bash
# round 2, call P: diagnose the LayerNorm-prologue numeric failure after the conversion change; rest of the suite without -x
set -u
OUT="gpurun_out/r02p"
mkdir -p "$OUT"
export TMPDIR=/tmp
./tools/diag/cvt_check > "$OUT/cvt_check.log" 2>&1; cat "$OUT/cvt_check.log"
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_serving.py -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|rel_err|AssertionError: assert" | head -40 > "$OUT/pytest_rest.log"
cat "$OUT/pytest_rest.log"
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -k "alternative_decode_pipelines or non_finite" 2>&1 | tail -15 > "$OUT/pytest_alt.log"
cat "$OUT/pytest_alt.log"
