#!/usr/bin/env bash
# the idle GPU time between the kernels of the decode loop: rocprofv3 kernel trace of a short bench + tools/step_gaps.py     bash tools/diag/step_gaps_run.sh [tag]
TAG="${1:-step_gaps}"; R=$PWD; OUT=gpurun_out/$TAG; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/trace -- python $R/bench.py --no-cpu-baseline --steps 1 --warmup 1 --new-tokens 200 --ttft-requests 0 > $R/$OUT/bench.log 2>&1 )
python tools/step_gaps.py $OUT/trace | tee $OUT/step_gaps.log
rm -rf $OUT/trace
