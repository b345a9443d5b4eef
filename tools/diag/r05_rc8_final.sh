OUT=gpurun_out/r05rc8; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_configs45.py tests/test_gpu_e2e.py -q -x -s -k "config4 or 8b_widths or one_launch_bit or 8b_full or 8b_dims" 2>&1 | grep -v amdgpu | grep -E "^\[|passed|failed|Error" | tail -14
timeout 300 python tools/ab_exp.py --new-tokens 512 --reps 2 16512 2>/dev/null | python -c "
import sys,json
print('1B us/step', [json.loads(l)['us_per_step'] for l in sys.stdin if l.startswith('{')])"
