OUT=gpurun_out/r05pk; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_long_context.py -q -s -rA -k "peaked" > $OUT/peaked.log 2>&1; echo rc=$?; grep -E "^\[|passed|failed|Error|assert" $OUT/peaked.log | tail -15
timeout 600 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_safety.py -q -x -k "one_launch or safety or tenant or poisoned" 2>&1 | tail -3
