# round 5: the default bench command a few times (an engine that owns its GPU: both fused decode launches), then the same with the row launches apart
run() { echo "=== $*"; env "$@" 2>&1 | grep -v amdgpu.ids | grep "^\[sv\]\|^{\|Error" | cut -c1-200; }
B="timeout 300 python bench.py --no-cpu-baseline"
run X=1 $B
run X=1 $B
run SV_EXP=8192 $B
