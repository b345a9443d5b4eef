for m in 131072 0 262144 524288; do
  echo "=== SV_EXP=$m (131072: old mt2 kernel; 0: LDS ring, depth by block count; 262144: CH=2 always; 524288: CH=4 where it fits)"
  SV_EXP=$m timeout 300 python bench.py --model 8b --weights fp8 --task text2svg --new-tokens 256 --steps 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('config5', d['value'], d['decode_us_per_step'], d['roofline']['avg_launch_us_gemm_chain'])"
  SV_EXP=$m timeout 300 python bench.py --beams 2 --sample --new-tokens 512 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('beam2  ', d['value'], d['decode_us_per_step'], d['roofline']['avg_launch_us_gemm_chain'])"
done
