run() { echo "=== $*"; env "$@" 2>&1 | grep -v amdgpu.ids | grep "^\[sv\]\|^{\|Error" | cut -c1-260; }
B="timeout 300 python bench.py --no-cpu-baseline --ttft-requests 1"
run X=1 $B --warmup 1 --steps 3
run X=1 $B --warmup 1 --steps 2
run X=1 $B --warmup 0 --steps 3
run SV_NO_GRAPH=1 $B --warmup 1 --steps 3
