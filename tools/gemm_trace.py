#!/usr/bin/env python3
"""Where a tile of the 256x256 big-M GEMM kernel spends its time (include/starvector_hip.h, sv_debug_gemm_trace): per block the
wall-clock stamps {start, K-tile 0 staged, K loop done, epilogue stored} of one launch on random operands, after 20 warm launches.
    python tools/gemm_trace.py [M N K act] ...        (default: the four decoder GEMMs of BASELINE config 2's prefill + 8192^3)
Prints, per shape: the launch span, the blocks per dispatch round, and min / median / p90 / max of every segment by round."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from starvector_amd import _lib  # noqa: E402

lib = _lib.load()
torch.zeros(1, device="cuda")
args = [int(x) for x in sys.argv[1:]]
shapes = [tuple(args[i:i + 4]) for i in range(0, len(args), 4)] or [
    (8288, 2304, 2048, 0), (8288, 2048, 2048, 0), (8288, 8192, 2048, 3), (8288, 2048, 8192, 0), (8192, 8192, 8192, 0)]


def pct(v):
    v = sorted(v)
    n = len(v)
    return [round(v[0], 2), round(v[n // 2], 2), round(v[min(n - 1, int(n * 0.9))], 2), round(v[-1], 2)]


for M, N, K, act in shapes:
    blocks = ((M + 255) // 256) * ((N + 255) // 256)
    buf = (C.c_int64 * (blocks * 16))()
    n = lib.sv_debug_gemm_trace(M, N, K, act, 1, buf, blocks)
    if n < 0:
        raise SystemExit(lib.sv_last_error().decode())
    rows = [list(buf[i * 8:(i + 1) * 8]) for i in range(n * 2)]
    rows = [r for r in rows if r[3]]
    t0 = min(r[0] for r in rows)
    us = lambda t: (t - t0) / 100.0
    span = max(us(r[3]) for r in rows)
    flops = 2.0 * M * N * K
    print(f"--- M {M} N {N} K {K} act {act}: {n} tiles of 256^2 ({n / 256:.2f} rounds of 256 CUs), K-tiles {K // 64}; launch span {span:.1f} us "
          f"= {flops / span / 1e6:.0f} TF/s")
    a = [r for r in rows if r[6] == 0]
    a.sort(key=lambda r: r[0])
    first = [r for r in a if us(r[0]) < 5.0]
    later = [r for r in a if us(r[0]) >= 5.0]
    for name, grp in (("blocks started in the first 5 us", first), ("blocks started later", later)):
        if not grp:
            continue
        print(f"  {name}: {len(grp)}")
        print(f"    start                         {pct([us(r[0]) for r in grp])}")
        print(f"    prologue (K-tile 0 staged)    {pct([(r[1] - r[0]) / 100.0 for r in grp])}")
        print(f"    K loop                        {pct([(r[2] - r[1]) / 100.0 for r in grp])}   per K-tile {pct([(r[2] - r[1]) / 100.0 / (K // 64) for r in grp])}")
        print(f"    epilogue (stored + drained)   {pct([(r[3] - r[2]) / 100.0 for r in grp])}")
        print(f"    end                           {pct([us(r[3]) for r in grp])}")
    b = {(r[4], r[5]): r for r in rows if r[6] == 7}
    skew = [(b[(r[4], r[5])][2] - r[2]) / 100.0 for r in a if (r[4], r[5]) in b]
    if skew:
        print(f"  group B (wave 7) leaves the K loop after group A (wave 0) by {pct(skew)} us")
    # idle between a CU's consecutive blocks cannot be seen per CU here; the chip-level view: blocks alive over time
    ev = sorted([(us(r[0]), 1) for r in a] + [(us(r[3]), -1) for r in a])
    alive, area, last = 0, 0.0, 0.0
    for t, d in ev:
        area += alive * (t - last)
        alive += d
        last = t
    print(f"  mean blocks alive over the span: {area / span:.1f} of 256")
