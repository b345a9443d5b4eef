#!/usr/bin/env python3
"""The compiler's per-kernel register / scratch / occupancy table at the current sources (no GPU needed):
    python tools/kernel_resources.py > profiles/kernel_resources_rNN.md"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "star-vector_amd"))
SRC = ["gemm", "decode_cols", "rowops", "attention", "sampling", "beam", "preprocess", "engine_core"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-kernarg-preload-count=16"]
print("# Kernel resources (hipcc -Rpass-analysis=kernel-resource-usage, gfx950, the build flags of star-vector_amd/build.py)\n")
print("| source | kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | static LDS B | occupancy waves/SIMD |")
print("|---|---|---|---|---|---|---|---|")
for f in SRC:
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null",
                        os.path.join(ROOT, "star-vector_amd", "csrc", f + ".hip")], capture_output=True, text=True)
    cur = {}
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
        else:
            cur[k.split(" ")[0]] = v
        if k.startswith("LDS Size"):
            dn = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
            dn = re.sub(r"\(.*", "", dn).replace("void ", "")
            print(f"| {f}.hip | `{dn}` | {cur.get('VGPRs')} | {cur.get('AGPRs')} | {cur.get('TotalSGPRs')} | {cur.get('ScratchSize')} | "
                  f"{cur.get('LDS')} | {cur.get('Occupancy')} |")
