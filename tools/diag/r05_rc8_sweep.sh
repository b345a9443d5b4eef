OUT=gpurun_out/r05rc8; mkdir -p $OUT
for d in ${DELAYS:-250 350 450 550 650}; do
  SV_RC_DELAY=$d timeout 300 python bench.py --model 8b --new-tokens 256 --steps 2 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg4 delay $d', d['value'], d.get('decode_us_per_step'))"
done
SV_EXP=8192 timeout 300 python bench.py --model 8b --new-tokens 256 --steps 2 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg4 off', d['value'], d.get('decode_us_per_step'))"
