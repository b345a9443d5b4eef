# wide rowln_cattn_kernel: activations in 5 + 4 (default build) against 3 + 3 + 3 (-DSV_RC_XB=3), config 4's workload, alternating builds on one box
run() { timeout 300 python bench.py --model 8b --new-tokens 256 --steps 2 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['value'], d.get('decode_us_per_step'))"; }
run "XB5"
SV_HIPCC_FLAGS=-DSV_RC_XB=3 python star-vector_amd/build.py > /dev/null 2>&1; run "XB3"
python star-vector_amd/build.py > /dev/null 2>&1; run "XB5"
SV_HIPCC_FLAGS=-DSV_RC_XB=3 python star-vector_amd/build.py > /dev/null 2>&1; run "XB3"
