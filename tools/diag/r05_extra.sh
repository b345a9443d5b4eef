# round 5 record: the other BASELINE workloads and the other lengths of config 2 on the final code
OUT=gpurun_out/r05extra; mkdir -p $OUT
timeout 300 python bench.py --no-cpu-baseline --new-tokens 256 > $OUT/bench_n1_256tok.json 2> $OUT/b256.err
timeout 400 python bench.py --no-cpu-baseline --new-tokens 4096 --steps 2 > $OUT/bench_n1_4096tok.json 2> $OUT/b4096.err
timeout 300 python bench.py --model 8b --new-tokens 256 --steps 2 --no-cpu-baseline > $OUT/bench_8b_im2svg.json 2> $OUT/b8b.err
timeout 300 python bench.py --model 8b --weights fp8 --task text2svg --new-tokens 256 --steps 2 --no-cpu-baseline > $OUT/bench_8b_fp8_text2svg.json 2> $OUT/b8bfp8.err
SV_DIST_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_2ranks_one_gpu_gloo.json 2> $OUT/b2.err
for f in $OUT/*.json; do echo "== $f"; python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(d['value'], d.get('ttft_p50_ms'), d.get('decode_us_per_step'), d['roofline']['frac'], d['roofline_whole_step']['frac'], d['config'].get('exclusive_device'))
except Exception as e: print('ERR', e)
PY
done
tail -3 $OUT/*.err | grep -v amdgpu | tail -12
