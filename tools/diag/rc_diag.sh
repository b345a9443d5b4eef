run() { echo "=== $*"; env "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-230; }
B="timeout 200 python bench.py --no-cpu-baseline --ttft-requests 1"
run X=1 $B --warmup 1 --steps 1
run X=1 $B --warmup 0 --steps 2
run SV_RC_POISON_KERNEL=1 $B --warmup 1 --steps 1
run SV_NO_GRAPH=1 $B --warmup 1 --steps 1
run SV_RC_DELAY=0 $B --warmup 1 --steps 1
run SV_EXP=1024 $B --warmup 1 --steps 1
run SV_EXP=512 $B --warmup 1 --steps 1
