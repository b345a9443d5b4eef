# first-poll delay of rowln_cattn_kernel (SV_RC_DELAY, 10 ns ticks) on the final code: us per decode step, 512 tokens, masks 128+16384
for d in 350 370 390 410 430; do
  echo "delay $d: $(SV_RC_DELAY=$d timeout 300 python tools/ab_exp.py --new-tokens 512 --reps 2 16512 2>/dev/null | python -c "
import sys,json
v=[json.loads(l)['us_per_step'] for l in sys.stdin if l.startswith('{')]
print(v)")"
done
