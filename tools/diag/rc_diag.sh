run() { echo "=== $*"; env "$@" 2>&1 | grep -v amdgpu.ids | grep "^\[sv\]\|^{\|Error" | cut -c1-330; }
B="timeout 300 python bench.py --no-cpu-baseline --ttft-requests 1"
run X=1 $B --warmup 1 --steps 2
run SV_EXP=1024 $B --warmup 1 --steps 2
run SV_EXP=8192 $B --warmup 1 --steps 2
