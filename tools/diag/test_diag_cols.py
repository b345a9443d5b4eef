"""Diagnosis only (not part of tests/): who moves when the in-block LayerNorm GEMM test fails after another test file --
the GPU result or the CPU reference?  Both are compared with a float64 CPU reference and a float32 GPU (rocBLAS) one."""
import pytest
import torch

from starvector_amd import engine as E
from tests.gpu_util import bf

pytestmark = pytest.mark.gpu


def test_who_moves():
    M, N, K = 32, 2304, 2048
    g = torch.Generator().manual_seed(11 * M + N + K)
    h = (1.5 * torch.randn(M, K, generator=g) + 0.3).bfloat16().float()
    gam = (1 + 0.1 * torch.randn(K, generator=g)).bfloat16().float()
    bet = (0.1 * torch.randn(K, generator=g)).bfloat16().float()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().float()
    b = (0.1 * torch.randn(N, generator=g)).bfloat16().float()
    ln32 = torch.nn.functional.layer_norm(h, (K,), gam, bet, 1e-5).bfloat16().float()
    ln64 = torch.nn.functional.layer_norm(h.double(), (K,), gam.double(), bet.double(), 1e-5).float().bfloat16().float()
    print("\n[diag] layer_norm fp32 vs fp64 (after bf16 rounding): differing elements", int((ln32 != ln64).sum()), "of", ln32.numel())
    ref32 = ln32 @ W.T + b
    ref64 = (ln32.double() @ W.double().T + b.double()).float()
    refgpu = (ln32.cuda() @ W.cuda().T + b.cuda()).cpu()
    got = E.op_decode_cols(bf(h), bf(W), bf(b), gamma=bf(gam), beta=bf(bet), out_f32=True).cpu()
    sc = float(ref64.abs().max())
    for name, t in (("cpu fp32 matmul", ref32), ("gpu fp32 matmul", refgpu), ("engine", got)):
        print(f"[diag] {name:16s} vs fp64 reference: max rel err {float((t - ref64).abs().max()) / sc:.3e}")
    print("[diag] torch threads", torch.get_num_threads(), "default dtype", torch.get_default_dtype(),
          "mkldnn", torch.backends.mkldnn.is_available(), getattr(torch.backends.mkldnn, "enabled", None))
    try:
        print("[diag] float32 matmul precision", torch.get_float32_matmul_precision())
    except Exception as e:       # noqa: BLE001
        print("[diag]", e)
