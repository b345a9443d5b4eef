# the frozen tree of the round: the whole GPU suite, smoke, the bench lines, then the same-box A/B against the round-4 tree (recipe for _ab_r04/: tools/diag/r05_bench_ab.sh)
bash tools/gpu_round_start.sh r05f4 pytestall smoke bench bench8b 2>&1 | tail -12
OUT=gpurun_out/r05f4; n=0
for tree in _ab_r04 . _ab_r04 .; do
  n=$((n+1)); tag=$( [ "$tree" = "." ] && echo r05 || echo r04 )_$n
  ( cd $tree && timeout 300 python bench.py --no-cpu-baseline ) > $OUT/ab_cfg2_${tag}.json 2> /dev/null
  ( cd $tree && timeout 300 python bench.py --model 8b --new-tokens 256 --steps 2 --no-cpu-baseline ) > $OUT/ab_cfg4_${tag}.json 2> /dev/null
done
for f in $OUT/ab_cfg*.json; do echo "== $f $(grep '^{' $f | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d.get('ttft_p50_ms'), d.get('decode_us_per_step'), d['roofline']['frac'], d['roofline_whole_step']['frac'])")"; done
