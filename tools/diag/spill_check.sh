#!/usr/bin/env bash
# Where does a kernel's spill code sit?  Compiles star-vector_amd/csrc/<file>.hip with -save-temps and lists, for the kernel whose mangled name contains
# <pattern>, every loop (label .. backward branch) that holds MFMAs together with the scratch instructions inside it.  A spill outside the K loop costs a
# few instructions per tile; inside it, it is a regression.   usage: tools/diag/spill_check.sh gemm gemm256_kernelIt
set -e
F="$1"; PAT="$2"; D=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c "star-vector_amd/csrc/$F.hip" -o "$D/o.o" -save-temps=obj 2>/dev/null
S=$(ls "$D"/*gfx950*.s | head -1)
python3 - "$S" "$PAT" <<'PY'
import re, sys
lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN\w*%s\w*:" % pat, l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
tot = sum("scratch_" in l for l in body)
print(f"{pat}: {len(body)} lines, {tot} scratch instructions")
for i, l in enumerate(body):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        seg = body[labels[m.group(1)]:i]
        mf = sum("v_mfma" in x for x in seg)
        sc = sum("scratch_" in x for x in seg)
        if mf:
            print(f"  loop {m.group(1)} lines {labels[m.group(1)]}-{i}: {mf} MFMAs, {sc} scratch instructions")
PY
rm -rf "$D"
