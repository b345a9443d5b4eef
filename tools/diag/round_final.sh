TAG="${1:-r06k}"
bash tools/gpu_round_start.sh ${TAG} pytest smoke bench bench8b rocprof pmc ctx 2>&1 | tail -30
OUT=gpurun_out/${TAG}
timeout 400 python bench.py --beams 2 --sample --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_beam2_n1.json 2> $OUT/beam2.err
timeout 300 python bench.py --no-cpu-baseline --new-tokens 256 > $OUT/bench_n1_256tok.json 2> $OUT/b256.err
timeout 400 python bench.py --no-cpu-baseline --new-tokens 4096 --steps 2 > $OUT/bench_n1_4096tok.json 2> $OUT/b4096.err
SV_DIST_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_2ranks_one_gpu_gloo.json 2> $OUT/b2.err
SV_GEMM_AUTOTUNE_LOG=1 timeout 300 python tools/ttft_ab.py --reps 3 0 2> $OUT/autotune.err > $OUT/ttft_one.log; grep "sv gemm autotune" $OUT/autotune.err | sort | uniq -c > $OUT/gemm_autotune_choices.log
timeout 300 python tools/prefill_ceiling.py > $OUT/prefill_ceiling.log 2>&1
for f in $OUT/bench_beam2_n1.json $OUT/bench_n1_256tok.json $OUT/bench_n1_4096tok.json $OUT/bench_2ranks_one_gpu_gloo.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1], d['value'], d.get('ttft_p50_ms'), d.get('decode_us_per_step'), d['roofline']['frac'], d['roofline_whole_step']['frac'])
except Exception as e: print('ERR', sys.argv[1], e)
PY
done
# third session: the gaps between the kernels of a decode step / of a request's prompt pass (rocprofv3 kernel traces)
R=$PWD
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/trace_steps -- python $R/bench.py --no-cpu-baseline --steps 1 --warmup 1 --new-tokens 200 --ttft-requests 0 > /dev/null 2>&1 )
python tools/step_gaps.py $OUT/trace_steps > $OUT/step_gaps.log 2>&1; rm -rf $OUT/trace_steps
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/trace_ttft -- python $R/tools/ttft_ab.py --reps 3 0 > /dev/null 2>&1 )
python tools/ttft_gaps.py $OUT/trace_ttft > $OUT/ttft_gaps.log 2>&1; rm -rf $OUT/trace_ttft
cat $OUT/step_gaps.log; head -3 $OUT/ttft_gaps.log
