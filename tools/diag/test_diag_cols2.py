"""Diagnosis only: the real test body of tests/test_gpu_ops.py::test_decode_cols_layernorm_prologue[32-2304-2048], then a look at
what differs when it fails."""
import pytest
import torch

from starvector_amd import engine as E
from tests.gpu_util import bf, rel_err

pytestmark = pytest.mark.gpu


def _ln_ref(x, w, b, eps=1e-5):
    return torch.nn.functional.layer_norm(x, (x.shape[-1],), w, b, eps).bfloat16().float()


def test_real_body():
    M, N, K = 32, 2304, 2048
    g = torch.Generator().manual_seed(11 * M + N + K)
    h = (1.5 * torch.randn(M, K, generator=g) + 0.3).bfloat16().float()
    gam = (1 + 0.1 * torch.randn(K, generator=g)).bfloat16().float()
    bet = (0.1 * torch.randn(K, generator=g)).bfloat16().float()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().float()
    b = (0.1 * torch.randn(N, generator=g)).bfloat16().float()
    ref = _ln_ref(h, gam, bet) @ W.T + b
    got = E.op_decode_cols(bf(h), bf(W), bf(b), gamma=bf(gam), beta=bf(bet), out_f32=True)
    e1 = rel_err(got, ref)
    got2 = E.op_decode_cols(bf(h), bf(W), bf(b), gamma=bf(gam), beta=bf(bet), out_f32=True)
    ref2 = _ln_ref(h, gam, bet) @ W.T + b
    ref64 = (_ln_ref(h, gam, bet).double() @ W.double().T + b.double()).float()
    sc = float(ref64.abs().max())
    print(f"\n[diag2] first call rel err vs first reference {e1:.3e}")
    print(f"[diag2] engine 1st vs 2nd launch identical: {torch.equal(got.cpu(), got2.cpu())}; reference 1st vs 2nd identical: {torch.equal(ref, ref2)}")
    for name, t in (("reference #1", ref), ("reference #2", ref2), ("engine #1", got.cpu()), ("engine #2", got2.cpu())):
        err = (t - ref64).abs()
        rows = (err.max(dim=1).values > 1e-3 * sc).nonzero().flatten().tolist()
        print(f"[diag2] {name:13s} vs fp64: max rel err {float(err.max()) / sc:.3e}, rows off: {rows}")
    # what kind of error is it?  a wrong row mean shifts every normalised value of the row by the same amount (error row
    # proportional to gamma . W^T); a wrong rstd scales them (error row proportional to the row's output minus beta . W^T - b)
    e = (got.cpu() - ref64)
    gW = gam @ W.T
    base = ref64 - (bet @ W.T + b)
    for r_ in (0, 5, 16, 17, 24, 31):
        er = e[r_]
        def corr(a, b_):
            return float((a * b_).sum() / (a.norm() * b_.norm() + 1e-30))
        print(f"[diag2] row {r_}: |err| {float(er.abs().max()):.3e}  corr with gamma.W^T {corr(er, gW):+.3f} (implied mean shift "
              f"{-float((er * gW).sum() / (gW * gW).sum()):+.4f} / rstd)  corr with scaled output {corr(er, base[r_]):+.3f} "
              f"(implied rstd factor {1 + float((er * base[r_]).sum() / (base[r_] * base[r_]).sum()):.5f})")
    hm = h.mean(-1)
    print("[diag2] true row means 16..19", hm[16:20].tolist(), "row stds", h.std(-1)[16:20].tolist())
