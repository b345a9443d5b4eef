"""-m gpu: every kernel of the path, called through the C ABI (sv_op_*), against a plain float32
restatement of the same operator on the same bf16-exact inputs.

Tolerances (stated, floating point): bf16 outputs must sit within ONE bf16 rounding of the float32
result -> |err| <= 2^-8 * max|ref| (plus the second rounding where the reference rounds an
intermediate); float32 outputs within 1e-5 relative (accumulation order only)."""
import os

import pytest
import torch

from starvector_amd import engine as E
from tests.gpu_util import bf, dev, rel_err, mean_err
from oracle import starvector_oracle as O

pytestmark = pytest.mark.gpu
BF16_1ULP = 2.0 ** -8


@pytest.mark.parametrize("M,D", [(1, 128), (5, 128), (257, 1024), (8288, 2048), (33, 4608)])
def test_layernorm_rows(M, D):
    g = torch.Generator().manual_seed(M * 7 + D)
    x = (2 * torch.randn(M, D, generator=g) + 0.5).bfloat16().float()
    w = (1 + 0.1 * torch.randn(D, generator=g)).bfloat16().float()
    b = (0.1 * torch.randn(D, generator=g)).bfloat16().float()
    ref = torch.nn.functional.layer_norm(x, (D,), w, b, 1e-5)
    got = E.op_layernorm(bf(x), bf(w), bf(b))
    assert rel_err(got, ref) <= 1.1 * BF16_1ULP and mean_err(got, ref) <= 5e-4


@pytest.mark.parametrize("M,N,K,act,res", [
    (1, 32, 64, "none", False), (32, 128, 64, "none", False), (300, 384, 128, "none", False),
    (512, 1024, 588, "none", False),            # patch-embed shape: K = 3*14*14, padded to 640 inside
    (257, 3072, 1024, "none", False),           # ViT in_proj
    (200, 512, 1024, "gelu_tanh", False), (130, 256, 256, "quickgelu", True), (1000, 2048, 1024, "swish", True),
    (129, 2304, 2048, "none", True),            # ragged M tile, N not a multiple of 128
    (64, 516, 256, "none", False),              # N % 32 != 0 (vocab-like)
])
def test_linear_mfma(M, N, K, act, res):
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).bfloat16().float()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().float()
    b = (0.1 * torch.randn(N, generator=g)).bfloat16().float()
    r = torch.randn(M, N, generator=g).bfloat16().float()
    y = x @ W.T + b
    if act != "none":
        yy = y.bfloat16().float()               # the reference rounds the Linear output before the activation
        y = {"gelu_tanh": lambda t: torch.nn.functional.gelu(t, approximate="tanh"),
             "quickgelu": lambda t: t * torch.sigmoid(1.702 * t),
             "swish": lambda t: t * torch.sigmoid(t)}[act](yy)
    if res:
        y = y.bfloat16().float() + r
    got = E.op_linear(bf(x), bf(W), bf(b), bf(r) if res else None, act=act)
    assert not torch.isnan(got.float()).any()
    assert rel_err(got, y) <= 2.2 * BF16_1ULP and mean_err(got, y) <= 6e-4


@pytest.mark.parametrize("M,N,K,act,res", [(2048 + 96, 512, 1024, "gelu_tanh", True),    # 128^2 tiles + tail kernel
                                           (2048 + 96, 8192, 2048, "none", True),       # 256^2 ping-pong tiles + tail
                                           (4096 + 32, 2048, 8192, "quickgelu", False), # long K through the tail kernel
                                           (2048 + 7, 1024, 640, "none", False)])       # ragged tail (7 rows), K = 10 * 64
def test_linear_big_m_kernels_agree_bitwise(M, N, K, act, res):
    """A big-M Linear is split into full 256-row tiles (128^2 or 256^2 kernel, picked by a cost model) and a remainder
    that goes through the one-wave-per-tile tail kernel.  All three accumulate K in the same order with the same MFMA,
    so a row's result must not depend on where it sits: rotate the rows so the tail rows land in full tiles and
    compare bit for bit (this is what keeps a request's tokens independent of the batch around it)."""
    g = torch.Generator().manual_seed(N + K)
    x = torch.randn(M, K, generator=g).bfloat16()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16()
    b = (0.1 * torch.randn(N, generator=g)).bfloat16()
    r = torch.randn(M, N, generator=g).bfloat16() if res else None
    tail = M % 256
    y = E.op_linear(bf(x), bf(W), bf(b), bf(r) if res else None, act=act).cpu()
    rot = lambda t: torch.cat([t[-tail:], t[:-tail]]).contiguous()
    y2 = E.op_linear(bf(rot(x)), bf(W), bf(b), bf(rot(r)) if res else None, act=act).cpu()
    assert torch.equal(y2.view(torch.int16), rot(y).view(torch.int16))
    ref = x.float() @ W.float().T + b.float()
    if act != "none":
        ref = {"gelu_tanh": lambda t: torch.nn.functional.gelu(t, approximate="tanh"),
               "quickgelu": lambda t: t * torch.sigmoid(1.702 * t)}[act](ref.bfloat16().float())
    if res:
        ref = ref.bfloat16().float() + r.float()
    assert rel_err(y, ref) <= 2.2 * BF16_1ULP and mean_err(y, ref) <= 6e-4


@pytest.mark.parametrize("M,N,K,act,res", [(8288, 2048, 2048, "none", True),             # prefill c_proj: 96 remainder rows, K = 8 chunks
                                           (8288, 2048, 8192, "none", True),             # down projection: K = 32 chunks (the steady loop)
                                           (8288, 8192, 2048, "gelu_tanh", False),       # c_fc: 256 column tiles
                                           (8224, 1024, 1024, "none", True),             # ViT out_proj: 32 remainder rows, K = 4 chunks (no steady state)
                                           (8224, 1024, 4096, "none", True),             # ViT MLP c_proj
                                           (512 + 70, 1000, 1280, "quickgelu", False),   # ragged rows and columns, K = 5 chunks
                                           (2048 + 33, 3072, 1536, "none", False),       # K = 6 chunks: one full pass of the 3-chunk loop
                                           (2048 + 96, 1024, 640, "none", False)])       # K = 640: the patch embedding's padded K
def test_remainder_row_kernels_agree_with_the_tile_kernels_bitwise(M, N, K, act, res):
    """The remainder rows through gemm_tail_kernel (one wave per 32 x 32 tile, FORCED: form 2 of sv_debug_set_gemm_form -- the tuned
    choice may or may not peel) against the 256^2 tile kernel computing the same rows as part of a 33rd tile row (form 1): the same
    MFMA in the same ascending k order -> the same bits.  (Written in round 5 for a four-wave form of the tail kernel that was
    bit-identical and slower -- profiles/gemm_tail4_r05_ab.log; the direct check of the peeled path stays.)"""
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).bfloat16()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16()
    b = (0.1 * torch.randn(N, generator=g)).bfloat16()
    r = torch.randn(M, N, generator=g).bfloat16() if res else None
    outs = {}
    try:
        for form in (1, 2):
            E.set_gemm_form(form)
            outs[form] = E.op_linear(bf(x), bf(W), bf(b), bf(r) if res else None, act=act).cpu()
    finally:
        E.set_gemm_form(-1)
    tail = M % 256
    for form in (2,):
        same = torch.equal(outs[form].view(torch.int16), outs[1].view(torch.int16))
        if not same:
            bad = (outs[form].view(torch.int16) != outs[1].view(torch.int16)).nonzero()
            raise AssertionError(f"form {form} != the 256^2 kernel at {bad.shape[0]} elements, first (row, col) {bad[0].tolist()} (rows >= {M - tail} are remainder rows)")
    ref = x.float() @ W.float().T + b.float()
    if act != "none":
        ref = {"gelu_tanh": lambda t: torch.nn.functional.gelu(t, approximate="tanh"),
               "quickgelu": lambda t: t * torch.sigmoid(1.702 * t)}[act](ref.bfloat16().float())
    if res:
        ref = ref.bfloat16().float() + r.float()
    assert rel_err(outs[2], ref) <= 2.2 * BF16_1ULP


@pytest.mark.parametrize("M,N,K,act,res", [(8288, 2304, 2048, "none", False),          # prefill c_attn: 297 tiles of 256^2
                                           (8288, 8192, 2048, "gelu_tanh", True),       # c_fc shape + residual: 1056 tiles
                                           (8224, 3072, 1024, "quickgelu", False),      # ViT: K = 16 K-tiles
                                           (4096 + 40, 1024 + 8, 512, "none", False)])  # ragged N and M, short K
def test_linear_tile_kernels_and_the_tuned_choice_agree_at_the_prefill_shapes(M, N, K, act, res):
    """The 128^2 kernel, the 256^2 kernel (rows not peeled) and whatever the tuner picks (peeled remainder rows through the tail kernel
    included) give the same bits at the shapes the prefill actually runs (forms through sv_debug_set_gemm_form)."""
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).bfloat16()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16()
    b = (0.1 * torch.randn(N, generator=g)).bfloat16()
    r = torch.randn(M, N, generator=g).bfloat16() if res else None
    outs = {}
    try:
        for form in (1, 0, -1):
            E.set_gemm_form(form)
            outs[form] = E.op_linear(bf(x), bf(W), bf(b), bf(r) if res else None, act=act).cpu()
    finally:
        E.set_gemm_form(-1)
    assert torch.equal(outs[1].view(torch.int16), outs[0].view(torch.int16))
    assert torch.equal(outs[1].view(torch.int16), outs[-1].view(torch.int16))
    ref = x.float() @ W.float().T + b.float()
    if act != "none":
        ref = {"gelu_tanh": lambda t: torch.nn.functional.gelu(t, approximate="tanh"),
               "quickgelu": lambda t: t * torch.sigmoid(1.702 * t)}[act](ref.bfloat16().float())
    if res:
        ref = ref.bfloat16().float() + r.float()
    assert rel_err(outs[1], ref) <= 2.2 * BF16_1ULP


@pytest.mark.parametrize("B,S,N,K,act,res", [(32, 259, 2048, 2048, "none", True),          # prefill c_proj: 32 tiles + 96 remainder rows
                                             (32, 259, 2048, 8192, "none", True),          # down projection: 16 chunks per wave
                                             (32, 259, 8192, 2048, "gelu_tanh", False),    # c_fc: NOT the per-sequence form (the cost model does not peel it): the structure changes nothing
                                             (32, 259, 2304, 2048, "none", False),         # c_attn: not either
                                             (32, 257, 1024, 4096, "none", True),          # ViT MLP c_proj: 1 row per image
                                             (32, 257, 1024, 1024, "none", True),          # ViT out_proj: 2 chunks per wave
                                             (5, 515, 1024, 640, "none", True),            # 2 tiles + 3 rows per sequence, 10 chunks over 8 waves
                                             (3, 258, 1000, 320, "quickgelu", False)])     # ragged N, 5 chunks: three waves have nothing to do
def test_linear_rows_a_sequence_leaves_over_its_tiles(B, S, N, K, act, res):
    """GemmArgs::seq_rows: where gemm_seq_form holds for the projection, the tile kernels cover the full 256-row tiles of EVERY sequence
    and gemm_tailk_kernel (K split over 8 waves) the S % 256 rows each sequence leaves over.  (a) against the fp32 reference; (b) rows
    inside full tiles carry the tile kernels' bits (the same call without the sequence structure); (c) a sequence's rows do not depend
    on the batch: the same sequence alone (B = 1) and at another place of the batch gives the same bits; (d) the compact form (the last
    rows of the sequences on their own: the pruned last prompt layer) gives those rows' bits; (e) where the form does not hold, the
    sequence structure changes nothing."""
    g = torch.Generator().manual_seed(B + S + N + K)
    M = B * S
    x = torch.randn(M, K, generator=g).bfloat16()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16()
    b = (0.1 * torch.randn(N, generator=g)).bfloat16()
    r = torch.randn(M, N, generator=g).bfloat16() if res else None
    rem = S % 256
    form = E.gemm_seq_form(S, N, K, act)
    try:
        E.set_linear_seq_rows(0)
        y_plain = E.op_linear(bf(x), bf(W), bf(b), bf(r) if res else None, act=act).cpu()
        E.set_linear_seq_rows(S)
        y = E.op_linear(bf(x), bf(W), bf(b), bf(r) if res else None, act=act).cpu()
        # (c) the last sequence alone, and the batch reversed sequence-wise
        xs, rs = x.view(B, S, K), (r.view(B, S, N) if res else None)
        y1 = E.op_linear(bf(xs[-1].contiguous()), bf(W), bf(b), bf(rs[-1].contiguous()) if res else None, act=act).cpu()
        yr = E.op_linear(bf(xs.flip(0).contiguous().view(M, K)), bf(W), bf(b), bf(rs.flip(0).contiguous().view(M, N)) if res else None, act=act).cpu()
        # (d) the last row of every sequence as a compact problem
        E.set_linear_seq_rows(-S)
        xc = xs[:, -1].contiguous()
        rc = rs[:, -1].contiguous() if res else None
        yc = E.op_linear(bf(xc), bf(W), bf(b), bf(rc) if res else None, act=act).cpu()
    finally:
        E.set_linear_seq_rows(0)
    ref = x.float() @ W.float().T + b.float()
    if act != "none":
        ref = {"gelu_tanh": lambda t: torch.nn.functional.gelu(t, approximate="tanh"),
               "quickgelu": lambda t: t * torch.sigmoid(1.702 * t)}[act](ref.bfloat16().float())
    if res:
        ref = ref.bfloat16().float() + r.float()
    assert rel_err(y, ref) <= 2.2 * BF16_1ULP and mean_err(y, ref) <= 6e-4
    yb, pb = y.view(B, S, N).view(torch.int16), y_plain.view(B, S, N).view(torch.int16)
    assert torch.equal(yb[:, :S - rem], pb[:, :S - rem])                 # every tile form sums the whole K in ascending order
    assert torch.equal(y1.view(torch.int16), yb[-1])
    assert torch.equal(yr.view(B, S, N).view(torch.int16), yb.flip(0))
    assert torch.equal(yc.view(torch.int16), yb[:, -1])
    if form:
        # the remainder rows are close to, not bit-equal with, the whole-K order (a different summation order, by design)
        assert not torch.equal(yb[:, S - rem:], pb[:, S - rem:])
        assert rel_err(y.view(B, S, N)[:, S - rem:], y_plain.view(B, S, N)[:, S - rem:].float()) <= 2.2 * BF16_1ULP
    else:
        assert torch.equal(yb, pb)
    print(f"[seq remainder] B={B} S={S} N={N} K={K}: per-sequence form {'ON' if form else 'off'}, {B * rem} remainder rows")


def test_linear_transpose_detecting():
    """A = identity with an ASYMMETRIC weight: catches swapped row/column in the MFMA C layout."""
    K = N = 128
    x = torch.eye(K)
    W = (torch.arange(N * K, dtype=torch.float32).view(N, K) % 251) / 16.0   # exact in bf16, W != W^T
    got = E.op_linear(bf(x), bf(W), None, None, out_f32=True)
    assert torch.equal(got.cpu(), W.T.contiguous())


def test_linear_f32_out():
    g = torch.Generator().manual_seed(9)
    x = torch.randn(70, 256, generator=g).bfloat16().float()
    W = (torch.randn(516, 256, generator=g) / 16).bfloat16().float()
    got = E.op_linear(bf(x), bf(W), None, None, out_f32=True)
    assert rel_err(got, x @ W.T) <= 1e-5


@pytest.mark.parametrize("M,N,K,sk", [(32, 64, 256, 1), (32, 2304, 2048, 4), (7, 516, 256, 2), (40, 2048, 8192, 4),
                                       (32, 96, 64, 1), (1, 8192, 2048, 1), (32, 49156, 2048, 1), (3, 256, 1024, 8)])
def test_linear_skinny_weight_streaming(M, N, K, sk):
    g = torch.Generator().manual_seed(M + N + K + sk)
    x = torch.randn(M, K, generator=g).bfloat16().float()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().float()
    b = (0.1 * torch.randn(N, generator=g)).bfloat16().float()
    got = E.op_linear_skinny(bf(x), bf(W), bf(b), splitk=sk)
    assert rel_err(got, x @ W.T + b) <= 1e-5
    # linearity (size-independent property): f(2x) == 2 f(x) exactly in fp32 up to accumulation order
    got2 = E.op_linear_skinny(bf(2 * x), bf(W), None, splitk=sk)
    got1 = E.op_linear_skinny(bf(x), bf(W), None, splitk=sk)
    assert torch.equal(got2, 2 * got1)
    # deterministic (fixed-order split-K reduction)
    assert torch.equal(got1, E.op_linear_skinny(bf(x), bf(W), None, splitk=sk))


@pytest.mark.parametrize("M,N,K,sk", [(64, 2304, 2048, 4), (33, 8192, 2048, 1), (64, 2048, 8192, 4), (50, 6144, 4608, 3),
                                       (64, 4608, 18432, 4), (40, 1024, 1024, 2)])
def test_linear_skinny_two_row_tiles_per_block_is_bitwise_the_one_tile_kernel(M, N, K, sk):
    """33..64 rows (BASELINE config 5's batch 64): one block feeds each weight fragment to both row tiles, so the weight stream
    crosses HBM once.  Same MFMA / k order / reduction order per row -> bit-identical to the one-tile-per-block kernels, bf16
    and fp8 weights."""
    g = torch.Generator().manual_seed(M + N + K + sk)
    x = torch.randn(M, K, generator=g).bfloat16().float()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().float()
    b = (0.1 * torch.randn(N, generator=g)).bfloat16().float()
    two = E.op_linear_skinny(bf(x), bf(W), bf(b), splitk=sk)
    two8, sc = E.op_linear_skinny_fp8(bf(x), bf(W), bf(b), splitk=sk)
    assert rel_err(two, x @ W.T + b) <= 1e-5
    # one, two or three column tiles per block (the engine picks per Linear; ragged last block when N / 32 is not a multiple):
    # the same bits every time
    # ... and whichever kernel streams them: both operands in registers (form 0, rounds 2-5) or the activations through a wave-private
    # LDS ring with hand-counted weight loads (round 6: forms 2 / 3 = two chunks of 2 / 4 k-steps per wave, form 1 = the launcher's pick)
    try:
        for form in (0, 2, 3, 1):
            E.set_skinny_form(form)
            for ct in (1, 2, 3):
                E.set_op_col_tiles(ct)
                assert torch.equal(E.op_linear_skinny(bf(x), bf(W), bf(b), splitk=sk), two), (form, ct)
                assert torch.equal(E.op_linear_skinny_fp8(bf(x), bf(W), bf(b), splitk=sk)[0], two8), (form, ct)
    finally:
        E.set_op_col_tiles(0)
        E.set_skinny_form(1)
    # each row tile alone (<= 32 rows: the one-tile-per-block kernel) gives the same bits as inside the two-tile launch: batch
    # composition cannot change a row
    for lo, hi in ((0, 32), (32, M)):
        assert torch.equal(E.op_linear_skinny(bf(x[lo:hi]), bf(W), bf(b), splitk=sk), two[lo:hi])
        assert torch.equal(E.op_linear_skinny_fp8(bf(x[lo:hi]), bf(W), bf(b), splitk=sk)[0], two8[lo:hi])


@pytest.mark.parametrize("M,N,K,act", [(32, 8192, 2048, "gelu_tanh"), (5, 512, 256, "none"), (40, 256, 1024, "gelu_tanh"),
                                        (3, 96, 64, "swish"), (64, 18432, 4608, "gelu_tanh"), (17, 40, 32, "quickgelu")])
def test_linear_skinny_fused_activation_epilogue(M, N, K, act):
    """The c_fc form of the decode GEMM: bias, round to bf16 (the reference's Linear output), activation, round -- in the kernel's
    epilogue, fragment-order output."""
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).bfloat16().float()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().float()
    b = (0.1 * torch.randn(N, generator=g)).bfloat16().float()
    y = (x @ W.T + b).bfloat16().float()
    y = {"gelu_tanh": lambda t: torch.nn.functional.gelu(t, approximate="tanh"), "swish": lambda t: t * torch.sigmoid(t),
         "quickgelu": lambda t: t * torch.sigmoid(1.702 * t), "none": lambda t: t}[act](y)
    got = E.op_linear_skinny_epi(bf(x), bf(W), bf(b), act=act)
    assert rel_err(got, y) <= 2.2 * BF16_1ULP and mean_err(got, y) <= 1.5e-3
    assert torch.equal(got, E.op_linear_skinny_epi(bf(x), bf(W), bf(b), act=act))


def test_linear_skinny_tail_tiles_split_along_k(monkeypatch):
    """StarVector-8B's c_fc at <= 32 rows is 576 column tiles on 512 block slots: the 64 tiles beyond the first round go to
    gemm_skinny_tailsplit_kernel (four K quarters per tile, fp32 partials, last arriver runs the epilogue).  The first 512 tiles are
    the one-tile kernel's bits; the split tiles sum in (quarter, wave) order: inside the same tolerance against the fp32 reference,
    the same bits at every batch <= 32, and the arrival tickets re-arm themselves (a second call gives the same bits).
    Reference op: /root/reference/starvector/model/llm/starcoder2.py:12-61 (the HF Starcoder2 MLP's c_fc + gelu_pytorch_tanh)."""
    M, N, K = 32, 18432, 4608
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("the split is sized on 2 x 256 block slots")
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).bfloat16().float()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().float()
    b = (0.1 * torch.randn(N, generator=g)).bfloat16().float()
    y = torch.nn.functional.gelu((x @ W.T + b).bfloat16().float(), approximate="tanh")
    monkeypatch.setenv("SV_TAILSPLIT", "0")
    one = E.op_linear_skinny_epi(bf(x), bf(W), bf(b), act="gelu_tanh")
    monkeypatch.setenv("SV_TAILSPLIT", "1")
    got = E.op_linear_skinny_epi(bf(x), bf(W), bf(b), act="gelu_tanh")
    first = 512 * 32
    assert torch.equal(got[:, :first], one[:, :first])
    assert not torch.equal(got[:, first:], one[:, first:]), "the split launch did not run (same bits as the one-tile kernel on 64 tiles)"
    assert rel_err(got, y) <= 2.2 * BF16_1ULP and mean_err(got, y) <= 1.5e-3
    assert rel_err(got[:, first:], one[:, first:].float()) <= 2.2 * BF16_1ULP
    assert torch.equal(got, E.op_linear_skinny_epi(bf(x), bf(W), bf(b), act="gelu_tanh"))
    for m in (1, 7):
        assert torch.equal(E.op_linear_skinny_epi(bf(x[:m]), bf(W), bf(b), act="gelu_tanh"), got[:m])


@pytest.mark.parametrize("M,V,K", [(32, 49156, 2048), (3, 1000, 256), (64, 49157, 4608)])
def test_linear_skinny_logits_epilogue(M, V, K):
    """The lm_head form: fp32 rows holding bf16-rounded values (HF casts the bf16 logits to float before the argmax)."""
    g = torch.Generator().manual_seed(M + V + K)
    x = torch.randn(M, K, generator=g).bfloat16().float()
    W = (torch.randn(V, K, generator=g) / K ** 0.5).bfloat16().float()
    got = E.op_linear_skinny_epi(bf(x), bf(W), out_f32=True).cpu()
    assert torch.equal(got, got.bfloat16().float())                      # every value is a bf16 value
    assert rel_err(got, x @ W.T) <= 1.1 * BF16_1ULP


@pytest.mark.parametrize("B,S,H,Hkv,hd,causal", [
    (2, 1, 2, 2, 64, 0), (2, 17, 2, 2, 64, 0), (2, 257, 16, 16, 64, 0),       # ViT MHSA shapes
    (3, 19, 2, 1, 128, 1), (2, 259, 16, 1, 128, 1),                           # decoder MQA prefill
    (1, 64, 4, 4, 128, 1), (1, 130, 4, 4, 128, 1), (1, 70, 8, 2, 64, 1), (1, 300, 16, 1, 128, 0),
    # more unmasked head_dim-64 shapes around the 128-row block edges, and the SigLIP tower's 577 rows
    (2, 129, 2, 2, 64, 0), (1, 160, 3, 3, 64, 0), (2, 384, 4, 4, 64, 0), (1, 385, 2, 2, 64, 0), (1, 577, 2, 2, 64, 0), (1, 128, 2, 2, 64, 0)])
def test_attention_prefill(B, S, H, Hkv, hd, causal):
    g = torch.Generator().manual_seed(B + S + H + hd)
    q = torch.randn(B, S, H * hd, generator=g).bfloat16().float()
    k = torch.randn(B, S, Hkv * hd, generator=g).bfloat16().float()
    v = torch.randn(B, S, Hkv * hd, generator=g).bfloat16().float()
    qq = q.view(B, S, H, hd).transpose(1, 2)
    kk = k.view(B, S, Hkv, hd).transpose(1, 2).repeat_interleave(H // Hkv, dim=1)
    vv = v.view(B, S, Hkv, hd).transpose(1, 2).repeat_interleave(H // Hkv, dim=1)
    s = qq @ kk.transpose(-1, -2) * hd ** -0.5
    if causal:
        s = s.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool), 1), float("-inf"))
    ref = (torch.softmax(s, -1) @ vv).transpose(1, 2).reshape(B, S, H * hd)
    got = E.op_attention(bf(q), bf(k), bf(v), H, Hkv, causal)
    assert not torch.isnan(got.float()).any()
    # probabilities are rounded to bf16 before P.V (flash-attn / HF eager both do): two roundings
    assert rel_err(got, ref) <= 3 * BF16_1ULP and mean_err(got, ref) <= 1e-3


def test_attention_online_softmax_rescale_branch():
    """Force the running max to jump at a later KV tile (rule: a rare data-dependent branch needs its
    own test): one key in the LAST tile dominates one query row."""
    B, S, H, hd = 1, 200, 2, 64
    g = torch.Generator().manual_seed(0)
    q = (0.1 * torch.randn(B, S, H * hd, generator=g))
    k = (0.1 * torch.randn(B, S, H * hd, generator=g))
    v = torch.randn(B, S, H * hd, generator=g)
    q[0, 150, :hd] = 4.0
    k[0, 190, :hd] = 4.0                          # score 4*4*64/8 = 128 >> all others, in KV tile 2
    q, k, v = q.bfloat16().float(), k.bfloat16().float(), v.bfloat16().float()
    qq = q.view(B, S, H, hd).transpose(1, 2); kk = k.view(B, S, H, hd).transpose(1, 2); vv = v.view(B, S, H, hd).transpose(1, 2)
    ref = (torch.softmax(qq @ kk.transpose(-1, -2) * hd ** -0.5, -1) @ vv).transpose(1, 2).reshape(B, S, H * hd)
    got = E.op_attention(bf(q), bf(k), bf(v), H, H, 0)
    assert rel_err(got, ref) <= 3 * BF16_1ULP
    torch.testing.assert_close(got.float().cpu()[0, 150, :hd], v[0, 190, :hd], rtol=0, atol=2e-2)


def test_plane_layernorm():
    g = torch.Generator().manual_seed(5)
    for (B, Q, D) in [(3, 17, 256), (2, 257, 2048)]:
        x = (torch.randn(B, Q, D, generator=g) + 0.3).bfloat16().float()
        w = (1 + 0.1 * torch.randn(Q, D, generator=g)).bfloat16().float()
        b = (0.1 * torch.randn(Q, D, generator=g)).bfloat16().float()
        ref = torch.nn.functional.layer_norm(x, (Q, D), w, b, 1e-5)
        got = E.op_plane_layernorm(bf(x), bf(w), bf(b))
        assert rel_err(got, ref) <= 1.1 * BF16_1ULP


def test_argmax_lowest_index_on_ties():
    g = torch.Generator().manual_seed(6)
    lg = torch.randn(33, 49156, generator=g)
    lg[2, 100] = lg[2, 40000] = 9.0
    lg[5, 49155] = 11.0
    lg[7] = 0.0                                   # all equal -> index 0
    got = E.op_argmax(lg.to(dev())).cpu().long()
    assert torch.equal(got, lg.argmax(-1))        # integer result: bit-exact
    assert got[2] == 100 and got[5] == 49155 and got[7] == 0


@pytest.mark.parametrize("V", [49156, 49157, 96])
def test_lm_head_with_folded_greedy_selection(V):
    """Round 5: the greedy arg-max rides in the lm_head launch's epilogue -- per block the best (value, lowest
    column) of its 32 columns, one 64-bit atomic max per row (gemm.hip, SkinnyArgs::amax) -- instead of argmax_kernel + a slice
    merge.  Integer result, bit-exact against torch.argmax of the kernel's own logits: exact ties (duplicated weight rows) resolve
    to the LOWEST index, the zero columns that pad the vocabulary to 32 never win (a row whose real logits are all negative), a
    NaN row has no winner, -0.0 ties with +0.0 like the float comparison does."""
    g = torch.Generator().manual_seed(60 + V)
    M, K = 32, 2048 if V > 1000 else 64
    x = torch.randn(M, K, generator=g).bfloat16()
    W = (torch.randn(V, K, generator=g) / K ** 0.5).bfloat16()
    W[V - 1] = W[7]                                   # an exact tie between column 7 and the last column (the 4 added tokens' tile)
    W[V // 2] = W[3]
    x[4] = x[4].abs()                                 # row 4: make column 9 the winner by a wide margin, and its duplicate V - 2 tie it
    W[9] = 0.25
    W[V - 2] = W[9]
    x[5] = -x[4]                                      # row 5: the same two columns are the most NEGATIVE; zero rows 11 / 12 give logits +-0.0
    W[11] = 0.0
    W[12] = -0.0
    x[6] = float("nan")                               # row 6: no comparable score at all
    lg, idx = E.op_lm_head_argmax(bf(x), bf(W))
    lg = lg.cpu()
    ok = [m for m in range(M) if m != 6]
    assert torch.equal(lg[ok], E.op_linear_skinny_epi(bf(x), bf(W), out_f32=True).cpu()[ok])        # the epilogue still stores the logits
    ref = lg[ok].argmax(-1)                           # torch: lowest index among equal maxima
    assert torch.equal(idx[ok], ref), (idx[ok], ref)
    assert int(idx[6]) == 0x7fffffff                  # what finish_step_kernel turns into token 0 + the device flag
    assert int(idx[4]) == 9 and float(lg[4, 9]) == float(lg[4, V - 2])
    # every real logit of a row negative: the zero columns that pad the vocabulary to a multiple of 32 (logit 0) are no candidates
    Wn = -W.abs()
    Wn[11] = Wn[10]; Wn[12] = Wn[13]                  # (no zero rows here: every real column is negative)
    xn = x.clone()
    xn[6] = x[7]
    xn[0] = x[0].abs()
    lg2, idx2 = E.op_lm_head_argmax(bf(xn), bf(Wn))
    lg2 = lg2.cpu()
    assert float(lg2[0].max()) < 0.0
    assert torch.equal(idx2, lg2.argmax(-1)) and int(idx2.max()) < V
    # +0.0 and -0.0 tie (the float comparison of argmax_kernel): rows of zeros in W, everything else negative -> the LOWER column
    Wz = Wn.clone()
    Wz[12] = 0.0
    Wz[11] = -0.0
    xz = xn.clone()
    xz[1] = -xn[0]                                    # row 1: 0 * negative = -0.0 in column 11 / 12 products; row 0: +0.0
    lg3, idx3 = E.op_lm_head_argmax(bf(xz), bf(Wz))
    assert int(idx3[0]) == 11 and float(lg3[0, 11]) == 0.0 and float(lg3[0, 12]) == 0.0
    assert torch.equal(idx3, lg3.cpu().argmax(-1))


@pytest.mark.parametrize("V,M", [(49156, 32), (49157, 32), (49156, 13), (49152, 1)])
def test_lm_head_persistent_blocks_bit_identical_to_one_tile_blocks(V, M):
    """Round 6 (fourth session): the lm_head of a <= 32-row step as one round of blocks that each walk several column tiles
    (gemm.hip gemm_head_persist_kernel: the wave's activation share in registers, the weights rolling through 16 registers per lane,
    the ragged last tile read column by column, one atomic max per row and BLOCK) against the 1537 one-tile blocks it replaces
    (SV_HEAD_PERSIST=0): logits and folded arg-max bit for bit, with ties, a NaN row and the vocabulary's ragged tile."""
    g = torch.Generator().manual_seed(90 + V + M)
    K = 2048
    x = torch.randn(M, K, generator=g).bfloat16()
    W = (torch.randn(V, K, generator=g) / K ** 0.5).bfloat16()
    W[V - 1] = W[7]                                   # a tie between column 7 and the last column (the ragged tile when V = 49156 / 49157)
    W[V - 3] = 0.5 * W[V - 3]
    if M > 6:
        x[6] = float("nan")
    outs = {}
    for mode in ("1", "0"):
        os.environ["SV_HEAD_PERSIST"] = mode
        try:
            lg, idx = E.op_lm_head_argmax(bf(x), bf(W))
            plain = E.op_linear_skinny_epi(bf(x), bf(W), out_f32=True)
            outs[mode] = (lg.cpu(), idx, plain.cpu())
        finally:
            del os.environ["SV_HEAD_PERSIST"]
    a, b = outs["1"], outs["0"]
    ok = [m for m in range(M) if m != 6]
    assert torch.equal(a[0][ok], b[0][ok]) and torch.equal(a[2][ok], b[2][ok]) and torch.equal(a[1], b[1])
    assert torch.equal(a[0][ok], a[2][ok])
    assert torch.equal(a[1][ok], a[0][ok].argmax(-1))
    if M > 6:
        assert bool(torch.isnan(a[0][6]).all()) and int(a[1][6]) == 0x7fffffff


@pytest.mark.parametrize("V,M", [(49156, 64), (49157, 40)])
def test_lm_head_persistent_blocks_two_row_tiles(V, M):
    """The same launch at 33..64 rows (the reference's default decode: 64 rows; gemm_head_persist_kernel<2>: both row tiles' activation shares in
    registers, every weight register feeds two MFMAs) against the two-row-tile LDS-ring kernel it replaces (SV_HEAD_PERSIST=0), bit for bit --
    and each row equal to what a one-row-tile launch gives it."""
    g = torch.Generator().manual_seed(95 + V + M)
    K = 2048
    x = torch.randn(M, K, generator=g).bfloat16()
    W = (torch.randn(V, K, generator=g) / K ** 0.5).bfloat16()
    outs = {}
    for mode in ("1", "0"):
        os.environ["SV_HEAD_PERSIST"] = mode
        try:
            outs[mode] = E.op_linear_skinny_epi(bf(x), bf(W), out_f32=True).cpu()
            if mode == "1":
                solo = torch.cat([E.op_linear_skinny_epi(bf(x[:32]), bf(W), out_f32=True).cpu(),
                                  E.op_linear_skinny_epi(bf(x[32:]), bf(W), out_f32=True).cpu()])
        finally:
            del os.environ["SV_HEAD_PERSIST"]
    assert outs["1"].shape == (M, V) and torch.isfinite(outs["1"]).all()
    assert torch.equal(outs["1"], outs["0"]) and torch.equal(outs["1"], solo)


def test_top_p_sampler_distribution():
    """Distributional parity with HF's temperature -> top-p -> multinomial (torch's RNG stream itself is
    not reproducible in a custom kernel)."""
    g = torch.Generator().manual_seed(7)
    V, n = 64, 20000
    lg = 2.0 * torch.randn(1, V, generator=g)
    probs = O.top_p_filtered_probs(lg, 0.8, 0.9)[0]
    rows = lg.repeat(n, 1).to(dev()).contiguous()
    s = E.op_sample_top_p(rows, 0.8, 0.9, seed=123, step=7).cpu().long()
    emp = torch.bincount(s, minlength=V).float() / n
    assert float(emp[probs == 0].sum()) == 0.0               # never samples outside the nucleus
    assert float((emp - probs).abs().sum()) < 0.03           # L1 distance (n = 20000 draws)
    # same (seed, step, row) -> same draw; different step -> different stream
    assert torch.equal(s, E.op_sample_top_p(rows, 0.8, 0.9, seed=123, step=7).cpu().long())
    assert not torch.equal(s, E.op_sample_top_p(rows, 0.8, 0.9, seed=123, step=8).cpu().long())
    # top_p -> tiny keeps only the argmax
    only = E.op_sample_top_p(rows[:64], 1.0, 1e-6, seed=1, step=0).cpu().long()
    assert bool((only == lg.argmax()).all())


def test_bf16_rounding_is_rne():
    """Both float->bf16 conversions used by the kernels (software f2bf in the epilogues, hardware
    v_cvt_pk_bf16_f32 in the LayerNorm prologue) must be round-to-nearest-EVEN like torch's cast:
    exact halfway cases decide it."""
    g = torch.Generator().manual_seed(11)
    x = torch.cat([torch.randn(4096, generator=g) * 3,
                   torch.tensor([1.0 + 2 ** -9, 1.0 + 2 ** -8 + 2 ** -9, -(1.0 + 2 ** -9), 2.0 + 2 ** -8, 0.0, 65280.0 * 2,
                                 1.0 + 2 ** -9 + 2 ** -20, 1.0 + 2 ** -9 - 2 ** -20, 3.0e-39])])
    ref = x.to(torch.bfloat16)
    got = E.op_cvt_bf16_hw(x.to(dev())).cpu()
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16))
    # software path: y = 1 * w + bias hits the halfway cases inside the GEMM epilogue
    w = torch.tensor([[1.0], [1.0 + 2 ** -7], [-1.0], [2.0]]).repeat(8, 1)          # N=32, K=1 (padded inside)
    W = torch.zeros(32, 64); W[:, :1] = w
    xx = torch.zeros(4, 64); xx[:, 0] = 1.0
    b = torch.full((32,), 2.0 ** -9)
    y = E.op_linear(bf(xx), bf(W), bf(b), None).float().cpu()
    expect = (xx @ W.T + b).to(torch.bfloat16).float()
    assert torch.equal(y, expect)


@pytest.mark.parametrize("w,h,c", [(224, 224, 3), (300, 300, 3), (100, 60, 4), (517, 333, 3), (64, 200, 4), (1024, 768, 3),
                                   (50, 50, 4), (223, 225, 4), (2048, 1536, 3), (1, 1, 3)])
def test_preprocess_image_bit_exact(w, h, c):
    """sv_preprocess_image == the reference's ImageTrainProcessor (Pillow paste / pad / BICUBIC resize + torchvision
    ToTensor / Normalize, restated with Pillow + torch here because torchvision is not installed), bit for bit: it is
    an integer / byte path up to the final two IEEE float operations."""
    import numpy as np
    from PIL import Image
    from oracle import image_preprocess as P
    rng = np.random.default_rng(w * 7 + h)
    px = rng.integers(0, 256, size=(h, w, c), dtype=np.uint8)
    if c == 4:
        px[..., 3] = rng.choice([0, 255, 128, 7, 254], size=(h, w))
    got = E.op_preprocess_image(torch.from_numpy(px).to(dev()), 224, P.CLIP_MEAN, P.CLIP_STD).cpu()
    img = Image.fromarray(px, "RGBA" if c == 4 else "RGB")
    if c == 4:
        bg = Image.new("RGB", img.size, (255, 255, 255))
        bg.paste(img, mask=img.split()[3])
        img = bg
    m = max(w, h)
    canvas = Image.new("RGB", (m, m), (255, 255, 255))
    canvas.paste(img, ((m - w) // 2, (m - h) // 2))
    if m != 224:
        canvas = canvas.resize((224, 224), Image.BICUBIC)
    ref = torch.from_numpy(np.asarray(canvas).copy()).permute(2, 0, 1).float().div(255.0)
    ref = (ref - torch.tensor(P.CLIP_MEAN).view(3, 1, 1)) / torch.tensor(P.CLIP_STD).view(3, 1, 1)
    assert got.shape == (3, 224, 224) and torch.equal(got.view(torch.int32), ref.view(torch.int32))
    assert np.array_equal(got.numpy().view(np.int32), P.preprocess(px).view(np.int32))            # and the numpy oracle
    # the mirror's processor takes the device path for RGB / RGBA and returns the same tensor
    from starvector_amd.model import ImageTrainProcessor
    pil = Image.fromarray(px, "RGBA" if c == 4 else "RGB")
    on_dev = ImageTrainProcessor(size=224, device=dev())(pil)
    on_host = ImageTrainProcessor(size=224)(pil)
    assert on_dev.is_cuda and torch.equal(on_dev.cpu().view(torch.int32), on_host.view(torch.int32))


@pytest.mark.parametrize("w,h,c", [(517, 300, 4), (384, 384, 3), (224, 224, 3), (1000, 120, 3), (384, 100, 4), (3, 5, 3)])
def test_preprocess_image_siglip_recipe_bit_exact(w, h, c):
    """Recipe 1 = HF SiglipImageProcessor (v2 tower): alpha dropped, stretch to 384 x 384, rescale in double, mean = std = 0.5.
    Bit-exact against the numpy oracle (pinned to HF's PIL processor on CPU) and, when importable here, against HF itself."""
    import numpy as np
    from PIL import Image
    from oracle import image_preprocess as P
    rng = np.random.default_rng(w + 13 * h)
    px = rng.integers(0, 256, size=(h, w, c), dtype=np.uint8)
    got = E.op_preprocess_image(torch.from_numpy(px).to(dev()), 384, (0.5,) * 3, (0.5,) * 3, recipe="siglip").cpu().numpy()
    assert np.array_equal(got.view(np.int32), P.preprocess_siglip(px).view(np.int32))
    from starvector_amd.model import SiglipProcessor
    pv = SiglipProcessor(384, dev())(images=[Image.fromarray(px, "RGBA" if c == 4 else "RGB")] * 2).pixel_values
    assert pv.shape == (2, 3, 384, 384) and np.array_equal(pv[1].cpu().numpy().view(np.int32), got.view(np.int32))


def test_preprocess_images_batched_stateless_bit_exact():
    """sv_preprocess_images: a batch of 37 images of mixed sizes / channel counts (copy-only, up- and down-scaling, RGBA) in
    ONE call (two chunks of <= 32: three launches each), tap tables computed on device -- every image bit-identical to the
    numpy restatement of Pillow + torchvision and to the single-image entry point; two interleaved callers on two streams do
    not disturb each other (no shared state: the workspace is the caller's)."""
    import numpy as np
    from PIL import Image
    from oracle import image_preprocess as P
    rng = np.random.default_rng(5)
    shapes = [(224, 224, 3), (224, 224, 4), (517, 300, 4), (64, 64, 3), (1000, 120, 3), (3, 5, 3), (1, 1, 4), (300, 517, 3),
              (2048, 1536, 3), (225, 224, 3)]
    imgs = []
    for i in range(37):
        w, h, c = shapes[i % len(shapes)]
        px = rng.integers(0, 256, size=(h, w, c), dtype=np.uint8)
        if c == 4:
            px[..., 3] = rng.choice([0, 255, 128, 7, 254], size=(h, w))
        imgs.append(px)
    dev_px = [torch.from_numpy(p).to(dev()) for p in imgs]
    got = E.op_preprocess_images(dev_px, 224, P.CLIP_MEAN, P.CLIP_STD)
    assert got.shape == (37, 3, 224, 224)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):                       # a second caller, concurrently, on its own stream and workspace
        other = E.op_preprocess_images(dev_px[::-1], 224, P.CLIP_MEAN, P.CLIP_STD)
    torch.cuda.synchronize()
    for i, px in enumerate(imgs):
        ref = P.preprocess(px)
        assert np.array_equal(got[i].cpu().numpy().view(np.int32), ref.view(np.int32)), i
        assert np.array_equal(other[36 - i].cpu().numpy().view(np.int32), ref.view(np.int32)), i
    one = E.op_preprocess_image(dev_px[2], 224, P.CLIP_MEAN, P.CLIP_STD)
    assert torch.equal(one.view(torch.int32), got[2].view(torch.int32))
    # the mirror's process_images batches the same way and keeps the reference's return shape (list of [1, 3, S, S])
    from starvector_amd.model import ImageTrainProcessor
    proc = ImageTrainProcessor(size=224, device=dev())
    pils = [Image.fromarray(px, "RGBA" if px.shape[2] == 4 else "RGB") for px in imgs[:5]]
    b = proc.batch(pils)
    assert b.shape == (5, 3, 224, 224) and torch.equal(b.view(torch.int32), got[:5].view(torch.int32))
    sig = E.op_preprocess_images(dev_px[:4], 384, (0.5,) * 3, (0.5,) * 3, recipe="siglip")
    for i in range(4):
        assert np.array_equal(sig[i].cpu().numpy().view(np.int32), P.preprocess_siglip(imgs[i]).view(np.int32)), i
    with pytest.raises(ValueError):
        E.op_preprocess_images([], 224, P.CLIP_MEAN, P.CLIP_STD)


@pytest.mark.parametrize("M,D,Kp,F", [(32, 2048, 2048, 8192), (5, 256, 256, 512), (17, 4608, 4608, 1024), (32, 128, 64, 96), (1, 512, 1024, 2048)])
def test_decode_output_projection_and_folded_layernorm(M, D, Kp, F):
    """The 6-launch decode layer (csrc/decode_cols.hip): the attention output projection over the whole K per block finishes the
    residual add in place (bit for bit the reference's bf16 cast points), and c_fc consumes the RAW residual stream with ln_2 folded
    into its weights and epilogue -- against LayerNorm -> Linear -> GELU in float32 with the reference's roundings."""
    g = torch.Generator().manual_seed(M + D + Kp + F)
    x = torch.randn(M, Kp, generator=g).bfloat16().float()
    Wp = (torch.randn(D, Kp, generator=g) / Kp ** 0.5).bfloat16().float()
    bp = (0.1 * torch.randn(D, generator=g)).bfloat16().float()
    h = (1.5 * torch.randn(M, D, generator=g) + 0.3).bfloat16().float()                 # a non-zero row mean: the fold subtracts mean * c1
    gam = (1 + 0.1 * torch.randn(D, generator=g)).bfloat16().float()
    bet = (0.1 * torch.randn(D, generator=g)).bfloat16().float()
    Wf = (torch.randn(F, D, generator=g) / D ** 0.5).bfloat16().float()
    bf_ = (0.1 * torch.randn(F, generator=g)).bfloat16().float()
    h2_ref = (h + (x @ Wp.T + bp).bfloat16().float()).bfloat16().float()
    h2, y = E.op_decode_proj_fold(bf(x), bf(Wp), bf(bp), bf(h), bf(gam), bf(bet), bf(Wf), bf(bf_))
    assert rel_err(h2, h2_ref) <= 1.1 * BF16_1ULP                                       # one bf16 rounding of the sum
    # c_fc against the reference chain evaluated on the ENGINE's h2 (so that only the fold is measured)
    h2e = h2.float().cpu()
    xln = torch.nn.functional.layer_norm(h2e, (D,), gam, bet, 1e-5)
    y_ref = torch.nn.functional.gelu((xln.bfloat16().float() @ Wf.T + bf_).bfloat16().float(), approximate="tanh")
    y_f32 = torch.nn.functional.gelu(xln @ Wf.T + bf_, approximate="tanh")               # no rounding of LN(h): what the fold approximates
    e_ref, e_f32 = rel_err(y, y_ref), rel_err(y, y_f32)
    assert min(e_ref, e_f32) <= 3 * BF16_1ULP and mean_err(y, y_f32) <= 2e-3, (e_ref, e_f32)
    h2b, yb = E.op_decode_proj_fold(bf(x), bf(Wp), bf(bp), bf(h), bf(gam), bf(bet), bf(Wf), bf(bf_))
    assert torch.equal(h2, h2b) and torch.equal(y, yb)                                   # deterministic (fixed-order statistics)


@pytest.mark.parametrize("ratio", [30.0, 100.0])
def test_folded_layernorm_with_a_large_common_offset_per_row(ratio):
    """the 6-launch layer applies ln_2 with single-pass statistics (var = E[x^2] - mean^2) and the epilogue
    rstd * (acc - mean * c1) + c2 -- both subtractions cancel when a row's |mean| is far above its standard deviation.  Rows with
    mean / std = 30 and 100 (the most a bf16 residual stream can carry: at 256 the bf16 grid spacing IS the standard deviation), both
    signs, against LayerNorm -> Linear -> GELU evaluated in float64 on the engine's own h2.  float32 loses (mean/std)^2 * 6e-8 of the
    variance (6e-4 at 100) and (mean/std) * 6e-8 of the projection: far below one bf16 rounding, which is what is asserted."""
    M, D, Kp, F = 32, 2048, 2048, 8192
    g = torch.Generator().manual_seed(int(ratio))
    x = torch.zeros(M, Kp)                                                               # h2 = h: the offset reaches c_fc unchanged
    Wp = (torch.randn(D, Kp, generator=g) / Kp ** 0.5).bfloat16().float()
    bp = torch.zeros(D)
    sign = torch.where(torch.arange(M) % 2 == 0, 1.0, -1.0)[:, None]
    h = (torch.randn(M, D, generator=g) + sign * ratio * (0.5 + torch.rand(M, 1, generator=g))).bfloat16().float()
    gam = (1 + 0.1 * torch.randn(D, generator=g)).bfloat16().float()
    bet = (0.1 * torch.randn(D, generator=g)).bfloat16().float()
    Wf = (torch.randn(F, D, generator=g) / D ** 0.5).bfloat16().float()
    bf_ = (0.1 * torch.randn(F, generator=g)).bfloat16().float()
    h2, y = E.op_decode_proj_fold(bf(x), bf(Wp), bf(bp), bf(h), bf(gam), bf(bet), bf(Wf), bf(bf_))
    assert torch.equal(h2.float().cpu(), h)
    h64 = h.double()
    got_ratio = float((h64.mean(1).abs() / h64.std(1)).min())
    assert got_ratio >= 0.45 * ratio
    xln = torch.nn.functional.layer_norm(h64, (D,), gam.double(), bet.double(), 1e-5)
    y64 = torch.nn.functional.gelu(xln @ Wf.double().T + bf_.double(), approximate="tanh").float()
    y_cast = torch.nn.functional.gelu((xln.float().bfloat16().float() @ Wf.T + bf_).bfloat16().float(), approximate="tanh")
    e64, ecast = rel_err(y, y64), rel_err(y, y_cast)
    print(f"[fold, row mean/std >= {got_ratio:.0f}] max err vs float64 LN-Linear-GELU {e64:.3e}, vs the bf16-cast-point chain {ecast:.3e}, "
          f"mean err {mean_err(y, y64):.3e}")
    assert min(e64, ecast) <= 3 * BF16_1ULP and mean_err(y, y64) <= 2e-3, (e64, ecast)
