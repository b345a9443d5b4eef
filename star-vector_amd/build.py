"""Build libstarvector_hip.so (gfx950) in-tree with hipcc.  No torch involved: the library is a
plain C-ABI shared object (include/starvector_hip.h)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libstarvector_hip.so")
SOURCES = ["gemm.hip", "decode_cols.hip", "rowops.hip", "attention.hip", "sampling.hip", "beam.hip", "preprocess.hip", "engine_core.hip", "engine_forward.hip", "engine_generate.hip", "engine_cb.hip", "engine_ops.hip"]
HEADERS = ["common.h", "kernels.h", "beam.h", "warp.h", "engine_internal.h", os.path.join("..", "..", "include", "starvector_hip.h"),
           os.path.join("..", "..", "include", "starvector_hip_debug.h")]
# -amdgpu-kernarg-preload-count: leading scalar / pointer kernel parameters arrive in SGPRs with the dispatch (gfx950) instead
# of through a scalar load at the top of every kernel (the decode step is 172 dependent launches)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"] + os.environ.get("SV_HIPCC_FLAGS", "").split()
if not os.environ.get("SV_NO_KERNARG_PRELOAD"):          # A/B switch of the build (profiles/kernarg_preload_r03_ab.log)
    FLAGS += ["-mllvm", "-amdgpu-kernarg-preload-count=16"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libstarvector_hip.so)")


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    # the flags are a dependency too: objects built with other SV_HIPCC_FLAGS / SV_NO_KERNARG_PRELOAD must not be reused
    stamp = os.path.join(OBJ, "flags.txt")
    flags_now = " ".join(FLAGS)
    if not os.path.exists(stamp) or open(stamp).read() != flags_now:
        force = True
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace(".hip", ".o"))
        if force or not _newer(obj, [src] + hdrs):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        return obj

    if jobs:
        if verbose:
            print(f"[starvector_amd.build] compiling {len(jobs)} HIP source(s) for gfx950", file=sys.stderr)
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(compile_one, jobs))
    with open(stamp, "w") as f:
        f.write(flags_now)
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or not _newer(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        if verbose:
            print(f"[starvector_amd.build] linked {LIB}", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
