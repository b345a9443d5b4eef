#!/usr/bin/env bash
# round 2, call Q: which ingredient makes the in-block LayerNorm GEMM (opt-in full-K pipeline) nondeterministic?
set -u
OUT="gpurun_out/r02q"
mkdir -p "$OUT"
export TMPDIR=/tmp
for v in 0 1 2 3; do
  touch star-vector_amd/csrc/decode_gemm.hip
  SV_HIPCC_FLAGS="-DDG_VARIANT=$v" python star-vector_amd/build.py > /dev/null 2>&1
  for rep in 1 2 3; do
    timeout 120 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "decode_cols_layernorm_prologue" 2>&1 | grep -E "passed|failed|AssertionError: assert" | tr '\n' ' ' | sed "s/^/variant $v rep $rep: /"
    echo
  done
done 2>&1 | tee "$OUT/cols_ln_variants.log"
