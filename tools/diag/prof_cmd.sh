#!/usr/bin/env bash
# rocprofv3 --kernel-trace of a command, summarised by (kernel, grid):  bash tools/diag/prof_cmd.sh <tag> <command ...>   -> gpurun_out/<tag>/by_grid.csv
TAG="$1"; shift
R=$PWD; OUT=gpurun_out/$TAG; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/trace -- "$@" > $R/$OUT/cmd.out 2> $R/$OUT/cmd.err )
python tools/trace_by_grid.py $OUT/trace $OUT/by_grid.csv > /dev/null 2>&1
rm -rf $OUT/trace
head -40 $OUT/by_grid.csv | cut -c1-150
