# same-box A/B: the round-4 tree against this tree, configs 2 / 4 / 5, alternating inside ONE gpurun call.
# Recipe for the other tree (not kept in the repository; _ab_r04/ is git-ignored but travels with the snapshot):
#   mkdir _ab_r04 && git archive <round-4 commit> | tar -x -C _ab_r04 && (cd _ab_r04 && python -c 'import __graft_entry__ as g; g.build()')
OUT=gpurun_out/r05ab2; mkdir -p $OUT; n=0
for tree in _ab_r04 . _ab_r04 .; do
  n=$((n+1)); tag=$( [ "$tree" = "." ] && echo r05 || echo r04 )_$n
  ( cd $tree && timeout 300 python bench.py --no-cpu-baseline ) > $OUT/cfg2_${tag}.json 2> $OUT/cfg2_${tag}.err
  ( cd $tree && timeout 300 python bench.py --model 8b --new-tokens 256 --steps 2 --no-cpu-baseline ) > $OUT/cfg4_${tag}.json 2> $OUT/cfg4_${tag}.err
  ( cd $tree && timeout 300 python bench.py --model 8b --weights fp8 --task text2svg --new-tokens 256 --steps 2 --no-cpu-baseline ) > $OUT/cfg5_${tag}.json 2> $OUT/cfg5_${tag}.err
done
for f in $OUT/cfg*.json; do echo "== $f"; grep "^{" $f | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d.get('ttft_p50_ms'), d.get('decode_us_per_step'), d['roofline']['frac'])" || tail -3 ${f%.json}.err; done
