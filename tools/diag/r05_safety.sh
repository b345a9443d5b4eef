OUT=gpurun_out/r05ab; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_safety.py -q -s -rA > $OUT/safety.log 2>&1; echo "safety rc=$?"; tail -25 $OUT/safety.log
