import os, sys, torch
sys.path.insert(0, "/root/repo")
from starvector_amd import engine as E
torch.zeros(1, device="cuda")
for N in (49152 - 32, 49152, 49156, 49152 + 32, 49152 + 64, 49152 + 256, 65536):
    us = min(E.bench_decode_linear(32, N, 2048, 1, 2, 200) for _ in range(3))
    mb = 2.0 * N * 2048 / 1e6
    print(f"lm_head N {N:6d} tiles {(N + 31) // 32:5d}: {us:7.2f} us  {mb / us:5.2f} TB/s", flush=True)
