OUT=gpurun_out/r05rc8; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -x -s -k "8b_widths or one_launch_bit" 2>&1 | grep -v amdgpu | tail -15
for m in 0 1; do
  if [ $m = 1 ]; then export SV_SHARED_GPU=; else export SV_EXP=8192; fi
  ( timeout 300 python bench.py --model 8b --new-tokens 256 --steps 2 --no-cpu-baseline ) > $OUT/cfg4_$m.json 2> $OUT/cfg4_$m.err
  unset SV_EXP
  grep "^{" $OUT/cfg4_$m.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg4 rc=$m', d['value'], d.get('decode_us_per_step'), d['roofline']['frac'])" || tail -3 $OUT/cfg4_$m.err
done
