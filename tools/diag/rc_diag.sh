# the default bench command, a few times (round 5: an intermittent give-up of a fused decode launch under the default steps / warmup)
for i in 1 2 3; do
  echo "=== run $i: default bench"
  timeout 200 python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-400
done
echo "=== SV_EXP=8192 (two row launches)"
SV_EXP=8192 timeout 200 python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-400
