"""CPU restatement of the StarVector-1B im2svg hot path (TEST INFRASTRUCTURE ONLY).

Every function cites the reference lines it follows (paths relative to
/root/reference).  The decoder arithmetic lives in a third-party dependency the
reference does not vendor at run time: ``transformers==4.49.0``
(pyproject.toml:18), classes ``GPTBigCodeForCausalLM`` + ``GenerationMixin``;
the dead in-tree copy ``starvector/model/gpt_bigcode/modeling_gpt_bigcode.py``
is the line-by-line source cited below.

Two precision modes:
  * ``mode="fp32"``  - the reference's CPU float32 path (BASELINE config 1).  This is the
    mode pinned against the reference modules by ``oracle/make_golden.py``.
  * ``mode="bf16"``  - same arithmetic in fp32, but every tensor the reference would hold
    in bf16 (``model_precision=bfloat16``: module outputs, residual stream, softmax
    probabilities, logits) is rounded to bf16 at that point.  GEMM / LayerNorm / softmax
    accumulate in fp32 exactly as ATen and flash-attn do (SURVEY.md section 8a "cast points").

Plain torch CPU ops only; no imports from the product package.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# state_dict prefixes (SURVEY.md section 8b "Weight names")
P_VIT = "model.image_encoder.visual_encoder."
P_LNV = "model.image_encoder.ln_vision."
P_ADP = "model.image_projection."
P_DEC = "model.svg_transformer.transformer.transformer."
K_LMH = "model.svg_transformer.transformer.lm_head.weight"


@dataclass
class OracleConfig:
    """Shapes of the path.  Defaults = StarVector-1B (SURVEY.md section 8 header)."""
    image_size: int = 224
    patch_size: int = 14
    vit_width: int = 1024
    vit_layers: int = 23          # image_encoder.py:52-58 (ViT-L/14 minus the last block)
    vit_heads: int = 16
    adapter_norm: str = "layer_norm"   # starvector_arch.py:103 ; "batch_norm" also supported
    hidden: int = 2048
    n_layer: int = 24
    n_head: int = 16
    n_inner: int = 8192
    vocab: int = 49156            # 49152 + [PAD] + 3 added tokens (llm/starcoder.py:40-53)
    n_positions: int = 8192
    eos_token_id: int = 0
    pad_token_id: int = 49152
    ln_eps: float = 1e-5
    # StarVector-8B (v2): SigLIP vision tower + StarCoder2 decoder
    arch: str = "v1"              # "v1": CLIP + GPTBigCode (MQA, learned positions); "v2": SigLIP + StarCoder2
    n_kv_head: int = 1            # v2: GQA key/value heads
    rope_theta: float = 1e6       # v2: rotary base (bigcode/starcoder2-7b config)
    vit_mlp: int = 4096           # v2: SigLIP intermediate size
    vit_eps: float = 1e-6         # v2: SigLIP layer_norm_eps
    sliding_window: int = 0       # v2: StarCoder2 attends to the last `sliding_window` keys (bigcode/starcoder2-7b: 4096); 0 = off

    @property
    def n_patches(self) -> int:
        return (self.image_size // self.patch_size) ** 2

    @property
    def query_length(self) -> int:   # starvector_base.py:85-106 (clip: 256 patches + cls; siglip_384: 576)
        return self.n_patches + (1 if self.arch == "v1" else 0)

    @property
    def head_dim(self) -> int:
        return self.hidden // self.n_head

    @staticmethod
    def starvector_8b() -> "OracleConfig":
        """StarVector-8B: siglip_384 (google/siglip-large-patch16-384) + bigcode/starcoder2-7b
        (configs/models/starvector-8b/im2svg-stack.yaml:7-11; llm/starcoder2.py:47 adds 4 tokens + [PAD])."""
        return OracleConfig(image_size=384, patch_size=16, vit_width=1024, vit_layers=24, vit_heads=16, vit_mlp=4096,
                            hidden=4608, n_layer=32, n_head=36, n_kv_head=4, n_inner=18432, vocab=49152 + 5,
                            n_positions=16384, eos_token_id=0, pad_token_id=0, arch="v2", sliding_window=4096)

    @staticmethod
    def tiny_v2() -> "OracleConfig":
        """Reduced v2 shapes (same op graph: SigLIP tower, RoPE, GQA with 3 query heads per KV head)."""
        return OracleConfig(image_size=64, patch_size=16, vit_width=128, vit_layers=2, vit_heads=2, vit_mlp=512,
                            hidden=768, n_layer=2, n_head=6, n_kv_head=2, n_inner=1024, vocab=517,
                            n_positions=256, eos_token_id=0, pad_token_id=0, arch="v2")

    @staticmethod
    def tiny() -> "OracleConfig":
        """Reduced shapes for golden fixtures / CPU tests (same op graph, same head dims)."""
        return OracleConfig(image_size=56, patch_size=14, vit_width=128, vit_layers=2, vit_heads=2,
                            hidden=256, n_layer=2, n_head=2, n_inner=1024, vocab=516,
                            n_positions=128, eos_token_id=0, pad_token_id=512)


# ----------------------------------------------------------------------------------------------
# precision model
# ----------------------------------------------------------------------------------------------
def _rounder(mode: str):
    if mode == "fp32":
        return lambda t: t
    if mode == "bf16":
        return lambda t: t.to(torch.bfloat16).to(torch.float32)
    if mode == "native":
        # the tensors ARE torch.bfloat16 (weights and inputs cast by the caller, any device): every op rounds where ATen
        # rounds -- a REAL bf16 execution of the same modules, used as the second comparator of the GPU parity tests
        # (SURVEY.md section 8c "run the same torch modules in bf16 on GPU when available")
        return lambda t: t
    raise ValueError(f"unknown mode {mode!r}")


def _patch_conv(image: Tensor, weight: Tensor, bias: Optional[Tensor], stride: int) -> Tensor:
    """Conv2d with kernel = stride = patch (clip_model.py:174; SigLIP patch_embedding).  In "native" bf16 mode on a GPU the
    convolution is written as the equivalent unfold + matmul so that the comparator does not depend on MIOpen's bf16
    convolution find step; float32 keeps F.conv2d (what the pinned goldens were minted with)."""
    if image.dtype == torch.float32:
        return F.conv2d(image, weight, bias, stride=stride)
    B, Cc, H, Wd = image.shape
    G = H // stride
    x = image.view(B, Cc, G, stride, G, stride).permute(0, 2, 4, 1, 3, 5).reshape(B, G * G, Cc * stride * stride)
    y = x @ weight.reshape(weight.shape[0], -1).T
    if bias is not None:
        y = y + bias
    return y.permute(0, 2, 1).reshape(B, weight.shape[0], G, G)


def _ln(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    # clip_model.py:117-124 (LayerNorm subclass computing in the weight dtype) and
    # gpt_bigcode/modeling_gpt_bigcode.py:676,680 (nn.LayerNorm, eps 1e-5); fp32 statistics.
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


# ----------------------------------------------------------------------------------------------
# deterministic weights + inputs (shared by tests, smoke and bench; values are bf16-exact so the
# fp32 oracle and the bf16 engine consume IDENTICAL numbers)
# ----------------------------------------------------------------------------------------------
def iter_weights(cfg: OracleConfig, seed: int = 1234, init: str = "parity", device=None):
    """Seeded random-init state_dict with the reference's key names and shapes, streamed as
    (name, float32 tensor) pairs so a loader never has to hold all 1.4 B parameters at once.

    init="parity": fan-in scaled normals so activations stay O(1) and greedy argmax has
    non-degenerate margins.  init="std002": N(0, 0.02) everywhere
    (throughput runs; values do not affect speed).  Values are bf16-exact.
    device: where the tensors are drawn (default: the host, which is what every committed fixture was minted with; a GPU
    generator draws OTHER values for the same seed -- used only where engine and oracle both consume the streamed tensors,
    e.g. the full-depth 8B parity test, to avoid minutes of host RNG).
    """
    g = torch.Generator(device=device).manual_seed(seed) if device is not None else torch.Generator().manual_seed(seed)

    def nrm(*shape, std):
        t = torch.empty(*shape, dtype=torch.float32, device=device).normal_(0.0, std, generator=g)
        return t.to(torch.bfloat16).to(torch.float32)

    def lin(name, out_f, in_f, gain=1.0):
        std = 0.02 if init == "std002" else gain / math.sqrt(in_f)
        yield name + ".weight", nrm(out_f, in_f, std=std)
        yield name + ".bias", nrm(out_f, std=0.02)

    def ln(name, *shape):
        if init == "std002":
            yield name + ".weight", torch.ones(*shape, device=device)
            yield name + ".bias", torch.zeros(*shape, device=device)
            return
        yield name + ".weight", (1.0 + 0.1 * torch.empty(*shape, device=device).normal_(0, 1, generator=g)).to(torch.bfloat16).to(torch.float32)
        yield name + ".bias", nrm(*shape, std=0.02)

    if cfg.arch == "v2":
        yield from _iter_weights_v2(cfg, init, nrm, lin, ln)
        return
    Dv, ps = cfg.vit_width, cfg.patch_size
    yield P_VIT + "conv1.weight", nrm(Dv, 3, ps, ps, std=(0.02 if init == "std002" else 1.0 / math.sqrt(3 * ps * ps)))
    yield P_VIT + "class_embedding", nrm(Dv, std=Dv ** -0.5 if init != "std002" else 0.02)
    yield P_VIT + "positional_embedding", nrm(cfg.query_length, Dv, std=(0.3 if init != "std002" else 0.02))
    yield from ln(P_VIT + "ln_pre", Dv)
    for i in range(cfg.vit_layers):
        p = f"{P_VIT}transformer.resblocks.{i}."
        yield from ln(p + "ln_1", Dv)
        for (nm, t) in lin(p + "attn.in_proj", 3 * Dv, Dv):
            yield nm.replace("in_proj.weight", "in_proj_weight").replace("in_proj.bias", "in_proj_bias"), t
        yield from lin(p + "attn.out_proj", Dv, Dv, gain=0.5)
        yield from ln(p + "ln_2", Dv)
        yield from lin(p + "mlp.c_fc", 4 * Dv, Dv)
        yield from lin(p + "mlp.c_proj", Dv, 4 * Dv, gain=0.5)
    yield from ln(P_LNV[:-1], Dv)

    D = cfg.hidden
    yield from lin(P_ADP + "c_fc", 2 * Dv, Dv)
    yield from lin(P_ADP + "c_proj", D, 2 * Dv)
    if cfg.adapter_norm == "layer_norm":
        yield from ln(P_ADP + "norm", cfg.query_length, D)
    else:
        Q = cfg.query_length
        yield from ln(P_ADP + "norm", Q)
        yield P_ADP + "norm.running_mean", nrm(Q, std=0.1)
        yield P_ADP + "norm.running_var", (1.0 + 0.2 * torch.rand(Q, generator=g, device=device)).to(torch.bfloat16).to(torch.float32)

    # wte is small (std 0.02): keeps the tied-head self-logit from dominating, so random-init greedy streams
    # are diverse instead of one repeated token
    yield P_DEC + "wte.weight", nrm(cfg.vocab, D, std=0.02)
    yield P_DEC + "wpe.weight", nrm(cfg.n_positions, D, std=(0.1 if init != "std002" else 0.02))
    kv = 2 * cfg.head_dim
    for i in range(cfg.n_layer):
        p = f"{P_DEC}h.{i}."
        yield from ln(p + "ln_1", D)
        yield from lin(p + "attn.c_attn", D + kv, D)
        yield from lin(p + "attn.c_proj", D, D, gain=0.5)
        yield from ln(p + "ln_2", D)
        yield from lin(p + "mlp.c_fc", cfg.n_inner, D)
        yield from lin(p + "mlp.c_proj", D, cfg.n_inner, gain=0.5)
    yield from ln(P_DEC + "ln_f", D)


P_DEC2 = "model.svg_transformer.transformer.model."      # Starcoder2ForCausalLM.model


def _iter_weights_v2(cfg: OracleConfig, init, nrm, lin, ln):
    """State-dict keys of the v2 model: HF SiglipVisionTransformer (image_encoder.py:41-43 keeps `.vision_model`)
    and HF Starcoder2ForCausalLM (llm/starcoder2.py:22-27)."""
    Dv, ps = cfg.vit_width, cfg.patch_size
    pe = P_VIT + "embeddings."
    yield pe + "patch_embedding.weight", nrm(Dv, 3, ps, ps, std=(0.02 if init == "std002" else 1.0 / math.sqrt(3 * ps * ps)))
    yield pe + "patch_embedding.bias", nrm(Dv, std=0.02)
    yield pe + "position_embedding.weight", nrm(cfg.n_patches, Dv, std=(0.3 if init != "std002" else 0.02))
    for i in range(cfg.vit_layers):
        p = f"{P_VIT}encoder.layers.{i}."
        yield from ln(p + "layer_norm1", Dv)
        yield from lin(p + "self_attn.q_proj", Dv, Dv)
        yield from lin(p + "self_attn.k_proj", Dv, Dv)
        yield from lin(p + "self_attn.v_proj", Dv, Dv)
        yield from lin(p + "self_attn.out_proj", Dv, Dv, gain=0.5)
        yield from ln(p + "layer_norm2", Dv)
        yield from lin(p + "mlp.fc1", cfg.vit_mlp, Dv)
        yield from lin(p + "mlp.fc2", Dv, cfg.vit_mlp, gain=0.5)
    yield from ln(P_VIT + "post_layernorm", Dv)
    D = cfg.hidden
    yield from lin(P_ADP + "c_fc", 2 * Dv, Dv)
    yield from lin(P_ADP + "c_proj", D, 2 * Dv)
    if cfg.adapter_norm == "layer_norm":
        yield from ln(P_ADP + "norm", cfg.query_length, D)
    else:
        raise NotImplementedError("v2 oracle weights: layer_norm adapter only (configs/models/starvector-8b)")
    yield P_DEC2 + "embed_tokens.weight", nrm(cfg.vocab, D, std=0.02)
    dh, kvd = cfg.head_dim, cfg.n_kv_head * cfg.head_dim
    for i in range(cfg.n_layer):
        p = f"{P_DEC2}layers.{i}."
        yield from ln(p + "input_layernorm", D)
        yield from lin(p + "self_attn.q_proj", cfg.n_head * dh, D)
        yield from lin(p + "self_attn.k_proj", kvd, D)
        yield from lin(p + "self_attn.v_proj", kvd, D)
        yield from lin(p + "self_attn.o_proj", D, cfg.n_head * dh, gain=0.5)
        yield from ln(p + "post_attention_layernorm", D)
        yield from lin(p + "mlp.c_fc", cfg.n_inner, D)
        yield from lin(p + "mlp.c_proj", D, cfg.n_inner, gain=0.5)
    yield from ln(P_DEC2 + "norm", D)


def fake_quantize_fp8(w: Dict[str, Tensor], cfg: OracleConfig) -> Dict[str, Tensor]:
    """The engine's `weight_dtype = fp8_e4m3` mode (BASELINE config 5's weight format; NOT a reference numerics mode --
    the reference reaches fp8 only through vLLM, which is not in tree): every decoder Linear weight and the lm_head is
    replaced by dequant(quant(W)): W taken as the bf16 tensor the engine is handed, one scale per output row
    = max|row| / 448, q = RNE to OCP e4m3.  Embedding lookups keep the bf16 table.  Returns a new dict."""
    out = dict(w)
    pre = P_DEC2 if cfg.arch == "v2" else P_DEC
    keys = [K_LMH]
    for i in range(cfg.n_layer):
        if cfg.arch == "v2":
            b = f"{pre}layers.{i}."
            keys += [b + n for n in ("self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight",
                                     "self_attn.o_proj.weight", "mlp.c_fc.weight", "mlp.c_proj.weight")]
        else:
            b = f"{pre}h.{i}."
            keys += [b + n for n in ("attn.c_attn.weight", "attn.c_proj.weight", "mlp.c_fc.weight", "mlp.c_proj.weight")]
    for k in keys:
        wb = w[k].to(torch.bfloat16).float()
        amax = wb.abs().amax(dim=1, keepdim=True)
        scale = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
        q = (wb / scale).to(torch.float8_e4m3fn).float()
        out[k] = q * scale
    return out


def embed_key(cfg: OracleConfig) -> str:
    return (P_DEC + "wte.weight") if cfg.arch == "v1" else (P_DEC2 + "embed_tokens.weight")


def make_weights(cfg: OracleConfig, seed: int = 1234, init: str = "parity") -> Dict[str, Tensor]:
    """The whole state_dict of iter_weights, plus the tied lm_head alias
    (gpt_bigcode/modeling_gpt_bigcode.py:1145; StarCoder2 ties embeddings too)."""
    w = dict(iter_weights(cfg, seed, init))
    w[K_LMH] = w[embed_key(cfg)]
    return w


def apply_fixture_weights(w: Dict[str, Tensor], cfg: OracleConfig, fixture: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """Fixtures minted by oracle/make_golden.py::fit_embedding carry a fitted tied embedding table ("wte", bf16): random-init
    transformers either repeat one token or decide by near-ties, so the table is fitted (a few hundred Adam steps on the
    oracle itself) until a designed, diverse token stream is the greedy one with a top-1/top-2 margin of a quarter of the
    logit scale -- far outside bf16 noise, so integer token parity is a hard assertion.  Returns a new dict."""
    if "wte" not in fixture:
        return w
    out = dict(w)
    t = fixture["wte"].to(torch.float32)
    out[embed_key(cfg)] = t
    out[K_LMH] = t
    return out


_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)   # data/util.py:33-38
_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def synthetic_images(batch: int, size: int = 224, seed: int = 0) -> Tensor:
    """Random-pixel images through CLIP normalisation, bf16-exact fp32."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(batch, 3, size, size, generator=g)
    mean = torch.tensor(_CLIP_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(_CLIP_STD).view(1, 3, 1, 1)
    return ((x - mean) / std).to(torch.bfloat16).to(torch.float32)


# ----------------------------------------------------------------------------------------------
# a2-a5: CLIP ViT + ln_vision
# ----------------------------------------------------------------------------------------------
def vit_forward(w: Dict[str, Tensor], cfg: OracleConfig, image: Tensor, mode: str = "fp32") -> Tensor:
    """clip_model.py:181-191 (VisionTransformer.forward) with blocks clip_model.py:130-155."""
    r = _rounder(mode)
    B = image.shape[0]
    Dv, H = cfg.vit_width, cfg.vit_heads
    dh = Dv // H
    # conv1, stride = kernel = patch, no bias (clip_model.py:174,182)
    x = _patch_conv(image, w[P_VIT + "conv1.weight"], None, cfg.patch_size)
    x = r(x.reshape(B, Dv, -1).permute(0, 2, 1))                       # :183-184
    cls = r(w[P_VIT + "class_embedding"]).expand(B, 1, Dv)              # :185
    x = torch.cat([cls, x], dim=1)
    x = r(x + w[P_VIT + "positional_embedding"])                        # :186
    x = r(_ln(x, w[P_VIT + "ln_pre.weight"], w[P_VIT + "ln_pre.bias"], cfg.ln_eps))   # :187
    scale = dh ** -0.5
    for i in range(cfg.vit_layers):
        p = f"{P_VIT}transformer.resblocks.{i}."
        # x = x + attn(ln_1(x))  (clip_model.py:148-153; nn.MultiheadAttention packed in_proj)
        h = r(_ln(x, w[p + "ln_1.weight"], w[p + "ln_1.bias"], cfg.ln_eps))
        qkv = r(h @ w[p + "attn.in_proj_weight"].T + w[p + "attn.in_proj_bias"])
        q, k, v = qkv.split(Dv, dim=-1)
        q = q.view(B, -1, H, dh).transpose(1, 2)
        k = k.view(B, -1, H, dh).transpose(1, 2)
        v = v.view(B, -1, H, dh).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) * scale                           # no mask (attn_mask=None)
        pr = r(torch.softmax(s, dim=-1))
        o = r((pr @ v).transpose(1, 2).reshape(B, -1, Dv))
        a = r(o @ w[p + "attn.out_proj.weight"].T + w[p + "attn.out_proj.bias"])
        x = r(x + a)
        # x = x + c_proj(QuickGELU(c_fc(ln_2(x))))  (clip_model.py:126-128,136-140,154)
        h = r(_ln(x, w[p + "ln_2.weight"], w[p + "ln_2.bias"], cfg.ln_eps))
        f = r(h @ w[p + "mlp.c_fc.weight"].T + w[p + "mlp.c_fc.bias"])
        f = r(f * torch.sigmoid(1.702 * f))
        m = r(f @ w[p + "mlp.c_proj.weight"].T + w[p + "mlp.c_proj.bias"])
        x = r(x + m)
    return x


def siglip_forward(w: Dict[str, Tensor], cfg: OracleConfig, image: Tensor, mode: str = "fp32") -> Tensor:
    """image_encoder.py:108-109 (siglip branch): HF SiglipVisionTransformer(...)["last_hidden_state"], i.e.
    patch conv (with bias) + learned positions, N pre-LN layers (MHA, GELU-tanh MLP), post_layernorm.
    Third-party arithmetic (transformers SiglipVisionTransformer; the reference pins transformers==4.49.0)."""
    r = _rounder(mode)
    B = image.shape[0]
    Dv, H = cfg.vit_width, cfg.vit_heads
    dh = Dv // H
    pe = P_VIT + "embeddings."
    x = _patch_conv(image, w[pe + "patch_embedding.weight"], w[pe + "patch_embedding.bias"], cfg.patch_size)
    x = r(x.flatten(2).transpose(1, 2))
    x = r(x + w[pe + "position_embedding.weight"])
    for i in range(cfg.vit_layers):
        p = f"{P_VIT}encoder.layers.{i}."
        h = r(_ln(x, w[p + "layer_norm1.weight"], w[p + "layer_norm1.bias"], cfg.vit_eps))
        q = r(h @ w[p + "self_attn.q_proj.weight"].T + w[p + "self_attn.q_proj.bias"]).view(B, -1, H, dh).transpose(1, 2)
        k = r(h @ w[p + "self_attn.k_proj.weight"].T + w[p + "self_attn.k_proj.bias"]).view(B, -1, H, dh).transpose(1, 2)
        v = r(h @ w[p + "self_attn.v_proj.weight"].T + w[p + "self_attn.v_proj.bias"]).view(B, -1, H, dh).transpose(1, 2)
        pr = r(torch.softmax((q @ k.transpose(-1, -2)) * dh ** -0.5, dim=-1))
        o = r((pr @ v).transpose(1, 2).reshape(B, -1, Dv))
        x = r(x + r(o @ w[p + "self_attn.out_proj.weight"].T + w[p + "self_attn.out_proj.bias"]))
        h = r(_ln(x, w[p + "layer_norm2.weight"], w[p + "layer_norm2.bias"], cfg.vit_eps))
        f = r(_gelu_tanh(r(h @ w[p + "mlp.fc1.weight"].T + w[p + "mlp.fc1.bias"])))
        x = r(x + r(f @ w[p + "mlp.fc2.weight"].T + w[p + "mlp.fc2.bias"]))
    return r(_ln(x, w[P_VIT + "post_layernorm.weight"], w[P_VIT + "post_layernorm.bias"], cfg.vit_eps))


def image_encoder_forward(w, cfg: OracleConfig, image: Tensor, mode: str = "fp32") -> Tensor:
    """image_encoder.py:91-94 (clip branch): ln_vision(visual_encoder(image)) on all tokens;
    image_encoder.py:108-109 (siglip branch) for v2."""
    if cfg.arch == "v2":
        return siglip_forward(w, cfg, image, mode)
    r = _rounder(mode)
    x = vit_forward(w, cfg, image, mode)
    return r(_ln(x, w[P_LNV + "weight"], w[P_LNV + "bias"], cfg.ln_eps))


# ----------------------------------------------------------------------------------------------
# a6: adapter
# ----------------------------------------------------------------------------------------------
def adapter_forward(w, cfg: OracleConfig, x: Tensor, mode: str = "fp32") -> Tensor:
    """adapter.py:33-39: dropout(eval: identity) -> c_fc -> Swish -> c_proj -> norm."""
    r = _rounder(mode)
    h = r(x @ w[P_ADP + "c_fc.weight"].T + w[P_ADP + "c_fc.bias"])
    h = r(h * torch.sigmoid(h))                                         # adapter.py:5-10
    h = r(h @ w[P_ADP + "c_proj.weight"].T + w[P_ADP + "c_proj.bias"])
    if cfg.adapter_norm == "layer_norm":
        # nn.LayerNorm([query_length, output_size]) : statistics over the joint plane, affine
        # weight/bias of shape [Q, D] (adapter.py:25-26)
        Q, D = h.shape[-2], h.shape[-1]
        h = F.layer_norm(h, (Q, D), w[P_ADP + "norm.weight"], w[P_ADP + "norm.bias"], cfg.ln_eps)
    else:
        # nn.BatchNorm1d(query_length) in eval: per-token affine from running stats (adapter.py:27-28)
        rm = w[P_ADP + "norm.running_mean"].view(1, -1, 1)
        rv = w[P_ADP + "norm.running_var"].view(1, -1, 1)
        h = (h - rm) / torch.sqrt(rv + cfg.ln_eps) * w[P_ADP + "norm.weight"].view(1, -1, 1) \
            + w[P_ADP + "norm.bias"].view(1, -1, 1)
    return r(h)


# ----------------------------------------------------------------------------------------------
# a7-a10: StarCoder (GPTBigCode, MQA) decoder
# ----------------------------------------------------------------------------------------------
def _gelu_tanh(x: Tensor) -> Tensor:
    # activation_function = gelu_pytorch_tanh (configuration_gpt_bigcode.py:107)
    return F.gelu(x, approximate="tanh")


def _block(w, cfg: OracleConfig, p: str, h: Tensor, k_cache: Optional[Tensor], v_cache: Optional[Tensor],
           r) -> Tuple[Tensor, Tensor, Tensor]:
    """GPTBigCodeBlock.forward (gpt_bigcode/modeling_gpt_bigcode.py:694-755) with MQA attention
    (:228-285, :151-226).  h: [B,S,D].  Returns (h_out, k_all [B,L,dh], v_all [B,L,dh])."""
    B, S, D = h.shape
    H, dh = cfg.n_head, cfg.head_dim
    x = r(_ln(h, w[p + "ln_1.weight"], w[p + "ln_1.bias"], cfg.ln_eps))
    qkv = r(x @ w[p + "attn.c_attn.weight"].T + w[p + "attn.c_attn.bias"])     # :138,253
    q, k, v = qkv.split((D, dh, dh), dim=-1)                                     # MQA: one kv head
    if k_cache is not None:                                                      # :265-269 cache append
        k = torch.cat([k_cache, k], dim=1)
        v = torch.cat([v_cache, v], dim=1)
    L = k.shape[1]
    q = q.view(B, S, H, dh).transpose(1, 2)                                      # [B,H,S,dh]
    s = torch.einsum("bhsd,bld->bhsl", q, k) * (dh ** -0.5)                      # scale_attn_weights
    # causal: query at absolute position L-S+i sees keys 0..L-S+i (all-ones padding mask)
    qi = torch.arange(L - S, L, device=h.device).view(S, 1)
    kj = torch.arange(L, device=h.device).view(1, L)
    s = s.masked_fill(kj > qi, float("-inf"))
    pr = r(torch.softmax(s, dim=-1))                                             # softmax in fp32 (:156-159)
    o = r(torch.einsum("bhsl,bld->bhsd", pr, v).transpose(1, 2).reshape(B, S, D))
    a = r(o @ w[p + "attn.c_proj.weight"].T + w[p + "attn.c_proj.bias"])
    h = r(h + a)
    x = r(_ln(h, w[p + "ln_2.weight"], w[p + "ln_2.bias"], cfg.ln_eps))
    f = r(x @ w[p + "mlp.c_fc.weight"].T + w[p + "mlp.c_fc.bias"])             # :645-660
    f = r(_gelu_tanh(f))
    m = r(f @ w[p + "mlp.c_proj.weight"].T + w[p + "mlp.c_proj.bias"])
    h = r(h + m)
    return h, k, v


def _lm_logits(w, cfg: OracleConfig, h_last: Tensor, r) -> Tensor:
    """ln_f + tied lm_head (gpt_bigcode/modeling_gpt_bigcode.py:1114,1258); logits are produced in
    model precision and cast to float32 before argmax/sampling (HF GenerationMixin._sample)."""
    x = r(_ln(h_last, w[P_DEC + "ln_f.weight"], w[P_DEC + "ln_f.bias"], cfg.ln_eps))
    return r(x @ w[K_LMH].T)


def _rope(cfg: OracleConfig, positions: Tensor, r):
    """Starcoder2RotaryEmbedding: inv_freq = theta^(-2i/d); cos/sin over cat(freqs, freqs), cast to the model dtype."""
    dh = cfg.head_dim
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, dh, 2, dtype=torch.float32, device=positions.device) / dh))
    fr = positions.to(torch.float32)[:, None] * inv[None, :]
    emb = torch.cat([fr, fr], dim=-1)
    return r(emb.cos()), r(emb.sin())


def _rot_half(x: Tensor) -> Tensor:
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], dim=-1)


def _block_v2(w, cfg: OracleConfig, p: str, h: Tensor, k_cache, v_cache, r):
    """Starcoder2DecoderLayer (transformers modeling_starcoder2): pre-LN, q/k/v/o projections with bias,
    rotary embedding (rotate_half convention), GQA, GELU-tanh MLP.  Sliding window: query i sees keys
    i - sliding_window < j <= i (HF's eager/sdpa mask and masking_utils.sliding_window_overlay; the KV cache itself is
    never cropped).  NOTE transformers 4.49's flash_attention_2 path -- the one the reference selects,
    llm/starcoder2.py:25 -- passes window_size=(W, W) to flash-attn, i.e. W + 1 keys; later releases pass W - 1
    ("we look at a max of `sliding_window` tokens back"), matching the mask used here.
    Returns (h, k_all [B,Hkv,L,dh], v_all)."""
    B, S, D = h.shape
    H, Hkv, dh = cfg.n_head, cfg.n_kv_head, cfg.head_dim
    x = r(_ln(h, w[p + "input_layernorm.weight"], w[p + "input_layernorm.bias"], cfg.ln_eps))
    q = r(x @ w[p + "self_attn.q_proj.weight"].T + w[p + "self_attn.q_proj.bias"]).view(B, S, H, dh).transpose(1, 2)
    k = r(x @ w[p + "self_attn.k_proj.weight"].T + w[p + "self_attn.k_proj.bias"]).view(B, S, Hkv, dh).transpose(1, 2)
    v = r(x @ w[p + "self_attn.v_proj.weight"].T + w[p + "self_attn.v_proj.bias"]).view(B, S, Hkv, dh).transpose(1, 2)
    past = 0 if k_cache is None else k_cache.shape[2]
    cos, sin = _rope(cfg, torch.arange(past, past + S, device=h.device), r)
    q = r(r(q * cos) + r(_rot_half(q) * sin))          # apply_rotary_pos_emb in model precision
    k = r(r(k * cos) + r(_rot_half(k) * sin))
    if k_cache is not None:
        k = torch.cat([k_cache, k], dim=2)
        v = torch.cat([v_cache, v], dim=2)
    L = k.shape[2]
    kk = k.repeat_interleave(H // Hkv, dim=1)
    vv = v.repeat_interleave(H // Hkv, dim=1)
    s = (q @ kk.transpose(-1, -2)) * dh ** -0.5
    qi = torch.arange(L - S, L, device=h.device).view(S, 1)
    kj = torch.arange(L, device=h.device).view(1, L)
    s = s.masked_fill(kj > qi, float("-inf"))
    if cfg.sliding_window:
        s = s.masked_fill(kj <= qi - cfg.sliding_window, float("-inf"))
    pr = r(torch.softmax(s, dim=-1))
    o = r((pr @ vv).transpose(1, 2).reshape(B, S, H * dh))
    h = r(h + r(o @ w[p + "self_attn.o_proj.weight"].T + w[p + "self_attn.o_proj.bias"]))
    x = r(_ln(h, w[p + "post_attention_layernorm.weight"], w[p + "post_attention_layernorm.bias"], cfg.ln_eps))
    f = r(_gelu_tanh(r(x @ w[p + "mlp.c_fc.weight"].T + w[p + "mlp.c_fc.bias"])))
    h = r(h + r(f @ w[p + "mlp.c_proj.weight"].T + w[p + "mlp.c_proj.bias"]))
    return h, k, v


def _forward_v2(w, cfg: OracleConfig, h: Tensor, cache, r):
    new_cache = []
    for i in range(cfg.n_layer):
        kc, vc = (None, None) if cache is None else cache[i]
        h, k, v = _block_v2(w, cfg, f"{P_DEC2}layers.{i}.", h, kc, vc, r)
        new_cache.append((k, v))
    x = r(_ln(h[:, -1, :], w[P_DEC2 + "norm.weight"], w[P_DEC2 + "norm.bias"], cfg.ln_eps))
    return r(x @ w[K_LMH].T), new_cache


def decoder_prefill(w, cfg: OracleConfig, inputs_embeds: Tensor, mode: str = "fp32"):
    """GPTBigCodeModel.forward over the 257+P prompt rows (gpt_bigcode/...:930-1134): position ids
    0..S0-1 from the all-ones mask (:980-985), hidden = inputs_embeds + wpe (:1060-1063).
    Returns (last-row logits [B,V] fp32, kv cache list[(k,v)])."""
    r = _rounder(mode)
    if cfg.arch == "v2":
        return _forward_v2(w, cfg, r(inputs_embeds), None, r)
    B, S0, D = inputs_embeds.shape
    h = r(inputs_embeds + w[P_DEC + "wpe.weight"][:S0])
    cache = []
    for i in range(cfg.n_layer):
        h, k, v = _block(w, cfg, f"{P_DEC}h.{i}.", h, None, None, r)
        cache.append((k, v))
    return _lm_logits(w, cfg, h[:, -1, :], r), cache


def decoder_forward_logits(w, cfg: OracleConfig, inputs_embeds: Tensor, num_logits_to_keep: int = 0, mode: str = "fp32"):
    """StarVectorForCausalLM.forward (starvector_arch.py:161-184): the decoder over inputs_embeds and the lm_head over the
    last `num_logits_to_keep` positions (0 = all).  v1 only (GPTBigCode); returns [B, n, V]."""
    assert cfg.arch == "v1"
    r = _rounder(mode)
    B, S, D = inputs_embeds.shape
    h = r(inputs_embeds + w[P_DEC + "wpe.weight"][:S])
    for i in range(cfg.n_layer):
        h, _, _ = _block(w, cfg, f"{P_DEC}h.{i}.", h, None, None, r)
    n = num_logits_to_keep if num_logits_to_keep and num_logits_to_keep > 0 else S
    return _lm_logits(w, cfg, h[:, -n:, :], r)


def decoder_decode_step(w, cfg: OracleConfig, tokens: Tensor, cache, mode: str = "fp32"):
    """One autoregressive step: wte[token] + wpe[pos] -> 24 blocks against the cache -> logits."""
    r = _rounder(mode)
    if cfg.arch == "v2":
        return _forward_v2(w, cfg, r(w[embed_key(cfg)][tokens]).unsqueeze(1), cache, r)
    pos = cache[0][0].shape[1]
    h = r(w[P_DEC + "wte.weight"][tokens] + w[P_DEC + "wpe.weight"][pos]).unsqueeze(1)
    new_cache = []
    for i in range(cfg.n_layer):
        h, k, v = _block(w, cfg, f"{P_DEC}h.{i}.", h, cache[i][0], cache[i][1], r)
        new_cache.append((k, v))
    return _lm_logits(w, cfg, h[:, -1, :], r), new_cache


# ----------------------------------------------------------------------------------------------
# a1, a11: generation orchestration
# ----------------------------------------------------------------------------------------------
def prepare_generation_inputs(w, cfg: OracleConfig, image: Tensor, prompt_ids: Tensor, mode: str = "fp32"):
    """starvector_base.py:203-221: encoder -> adapter -> cat(visual, wte(prompt_ids)); ones mask."""
    r = _rounder(mode)
    vis = adapter_forward(w, cfg, image_encoder_forward(w, cfg, image, mode), mode)
    tok = r(w[embed_key(cfg)][prompt_ids])                               # starvector_v1.py:16-18 / starvector_v2.py:45-47
    return torch.cat([vis, tok], dim=1)


def greedy_generate(w, cfg: OracleConfig, inputs_embeds: Tensor, max_length: int,
                    stop_ids: Optional[Sequence[int]] = None, mode: str = "fp32",
                    return_logits: bool = False, repetition_penalty: float = 1.0, min_length: int = 0):
    """HF GenerationMixin.generate -> _sample with do_sample=False, num_beams=1, as driven by
    starvector_base.py:228-241,255.  Semantics:
      * with inputs_embeds the new-token budget is max_length - S0;
      * only NEW tokens are returned;
      * finished rows (EOS seen) emit pad_token_id;
      * StoppingCriteriaSub (starvector_base.py:9-20) looks at ROW 0 only and stops the batch;
      * generation ends when every row is finished, the stop fires, or the budget is spent;
      * repetition_penalty (starvector_base.py:237): HF RepetitionPenaltyLogitsProcessor over the ids generated
        so far (with inputs_embeds the prompt has no ids): score < 0 ? score * p : score / p;
      * min_length (starvector_base.py:236, default 1; validation configs pass 10): with inputs_embeds HF first
        subtracts the prompt length (GenerationMixin._prepare_generated_length: min_length = max(min_length - S0, 0)),
        then MinLengthLogitsProcessor sets the EOS score to -inf while fewer than that many tokens have been
        generated.  On the im2svg path S0 >= 258 makes it 0; a short text2svg caption can leave it positive.
    """
    B, S0, _ = inputs_embeds.shape
    min_new = max(int(min_length) - S0, 0)
    budget = max_length - S0
    if budget <= 0:
        raise ValueError("max_length must exceed the prompt length (HF raises here)")
    logits, cache = decoder_prefill(w, cfg, inputs_embeds, mode)
    unfinished = torch.ones(B, dtype=torch.bool, device=inputs_embeds.device)
    out: List[Tensor] = []
    all_logits: List[Tensor] = []
    stop = list(stop_ids) if stop_ids else None
    for t in range(budget):
        scores = logits.float()
        if repetition_penalty != 1.0 and out:
            prev = torch.stack(out, dim=1)
            g = torch.gather(scores, 1, prev)
            g = torch.where(g < 0, g * repetition_penalty, g / repetition_penalty)
            scores = scores.scatter(1, prev, g)
        if t < min_new and 0 <= cfg.eos_token_id < scores.shape[1]:
            scores = scores.clone()
            scores[:, cfg.eos_token_id] = -float("inf")
        if return_logits:
            all_logits.append(scores)
        nxt = torch.argmax(scores, dim=-1)
        nxt = torch.where(unfinished, nxt, torch.full_like(nxt, cfg.pad_token_id))
        out.append(nxt)
        unfinished = unfinished & (nxt != cfg.eos_token_id)
        fired = False
        if stop is not None:
            row0 = [int(o[0]) for o in out[-len(stop):]]
            fired = row0 == stop
        if fired or not bool(unfinished.any()) or t == budget - 1:
            break
        logits, cache = decoder_decode_step(w, cfg, nxt, cache, mode)
    toks = torch.stack(out, dim=1)
    if return_logits:
        return toks, torch.stack(all_logits, dim=1)
    return toks


class BeamSearchState:
    """HF GenerationMixin._beam_search (do_sample=False) as the reference drives it with its default num_beams=2
    (starvector_base.py:234,238,293): the bookkeeping between two forward passes, restated from the algorithm.
    Per request b (rows b*num_beams .. of the expanded batch):
      * running scores start [0, -1e9, ...];
      * each step: log_softmax (fp32) -> repetition penalty on the LOG-PROBS of each beam's own ids -> add the beam's
        running score -> the K = 2*num_beams best (beam, token) continuations over num_beams*V, best first;
        with do_sample (the reference's training-time validation uses num_beams 3 + nucleus sampling,
        configs/models/starvector-8b/im2svg-stack.yaml:75-81) the log-probs first go through the warpers and the K
        continuations are DRAWN without replacement from softmax(accumulated scores), kept in draw order;
      * a continuation "hits" when its token is EOS, when it reaches the budget, or when the reference's row-0
        StoppingCriteriaSub fires (it looks at the best continuation of request 0 only and returns a plain bool, which
        HF ORs into every row of every request);
      * the num_beams best non-hitting continuations run on (hitting ones get -1e9 added);
      * hitting continuations ranked inside the first num_beams compete, with score / len**length_penalty, for the
        num_beams finished slots (blocked once the request is full under early_stopping=True, or once its heuristic said
        no improvement is possible);
      * the loop ends when no request can improve, when (early_stopping=True) all finished slots of all requests are
        full, or when every continuation hit;
      * result(): the best finished hypothesis per request, cropped to the longest, filled with pad (or eos when pad is
        0 / unset -- HF's `pad or eos`)."""

    def __init__(self, batch: int, num_beams: int, vocab: int, budget: int, eos_token_id: int, pad_token_id: int,
                 length_penalty: float = 1.0, early_stopping=True, stop_ids: Optional[Sequence[int]] = None,
                 repetition_penalty: float = 1.0, do_sample: bool = False, temperature: float = 1.0,
                 top_p: float = 1.0, top_k: int = 0, min_new_tokens: int = 0):
        self.sample = (bool(do_sample), temperature, top_p, top_k)
        self.min_new = int(min_new_tokens)       # MinLengthLogitsProcessor, applied to the LOG-PROBS (after the penalty)
        self.B, self.nb, self.V, self.budget = batch, int(num_beams), vocab, budget
        self.K = 2 * self.nb
        self.eos, self.lp, self.es, self.pen = eos_token_id, length_penalty, early_stopping, repetition_penalty
        self.fill = pad_token_id if pad_token_id else eos_token_id
        self.stop = list(stop_ids) if stop_ids else None
        B, nb = self.B, self.nb
        self.run_seq = torch.full((B, nb, budget), self.fill, dtype=torch.long)
        self.run_score = torch.zeros(B, nb, dtype=torch.float32)
        self.run_score[:, 1:] = -1.0e9
        self.fin_seq = self.run_seq.clone()
        self.fin_len = torch.zeros(B, nb, dtype=torch.long)
        self.fin_score = torch.full((B, nb), -1.0e9, dtype=torch.float32)
        self.fin_done = torch.zeros(B, nb, dtype=torch.bool)
        self.can_improve = torch.ones(B, dtype=torch.bool)
        self.cur = 0

    def step(self, logits: Tensor):
        """logits [B*nb, V] of the current running beams -> (go_on, flat parent rows [B*nb], tokens [B*nb])."""
        B, nb, V, K, cur = self.B, self.nb, self.V, self.K, self.cur
        NEG = torch.tensor(-1.0e9, dtype=torch.float32)
        ar_b = torch.arange(B)[:, None]
        lp = torch.log_softmax(logits.float(), dim=-1)
        if self.pen != 1.0 and cur > 0:
            prev = self.run_seq[:, :, :cur].reshape(B * nb, cur)
            g = torch.gather(lp, 1, prev)
            g = torch.where(g < 0, g * self.pen, g / self.pen)
            lp = lp.scatter(1, prev, g)
        if cur < self.min_new and 0 <= self.eos < V:
            lp = lp.clone()
            lp[:, self.eos] = -float("inf")
        if self.sample[0]:                                                # beam-sample: warpers act on the log-probs
            lp = warp_scores(lp, self.sample[1], self.sample[2], self.sample[3], min_tokens_to_keep=2)
        acc = (lp.view(B, nb, V) + self.run_score[:, :, None]).reshape(B, nb * V)
        if self.sample[0]:                                                # K draws without replacement, in draw order
            c_idx = torch.multinomial(torch.softmax(acc, dim=-1), num_samples=K)
            c_score = torch.gather(acc, 1, c_idx)
        else:
            c_score, c_idx = torch.topk(acc, K, dim=1)                   # best first
        c_beam, c_tok = c_idx // V, c_idx % V
        c_seq = self.run_seq[ar_b, c_beam]                                # [B, K, budget]
        c_seq[:, :, cur] = c_tok
        hits = (c_tok == self.eos) | (cur + 1 >= self.budget)
        st = self.stop
        if st is not None and cur + 1 >= len(st) and c_seq[0, 0, cur + 1 - len(st):cur + 1].tolist() == st:
            hits = torch.ones_like(hits)
        # running beams for the next step
        r_score = c_score + hits.float() * NEG
        sel = torch.topk(r_score, nb, dim=1)[1]
        self.run_seq = c_seq[ar_b, sel]
        self.run_score = r_score[ar_b, sel]
        parent = c_beam[ar_b, sel]
        # finished slots
        just = hits & (torch.arange(K) < nb)[None, :]
        f = c_score / float((cur + 1) ** self.lp)
        full = self.fin_done.all(dim=1, keepdim=True) & (self.es is True)
        f = f + full.float() * NEG
        f = f + (~self.can_improve)[:, None].float() * NEG
        f = f + (~just).float() * NEG
        m_score = torch.cat([self.fin_score, f], dim=1)
        m_sel = torch.topk(m_score, nb, dim=1)[1]
        self.fin_score = m_score[ar_b, m_sel]
        self.fin_seq = torch.cat([self.fin_seq, c_seq], dim=1)[ar_b, m_sel]
        self.fin_len = torch.cat([self.fin_len, torch.full((B, K), cur + 1)], dim=1)[ar_b, m_sel]
        self.fin_done = torch.cat([self.fin_done, just], dim=1)[ar_b, m_sel]
        self.cur = cur = cur + 1
        # can the running beams still beat the worst finished hypothesis?
        hyp_len = self.budget if (self.es == "never" and self.lp > 0.0) else cur
        best_run = self.run_score[:, :1] / float(hyp_len ** self.lp)
        worst = torch.where(self.fin_done, self.fin_score.min(dim=1, keepdim=True)[0], NEG)
        self.can_improve = self.can_improve & (best_run > worst).any(dim=1)
        go_on = bool(self.can_improve.any()) and not (bool(self.fin_done.all()) and self.es is True) \
            and not bool(hits.all())
        return go_on, (parent + ar_b * nb).reshape(-1), self.run_seq[:, :, cur - 1].reshape(-1)

    def result(self):
        L = int(self.fin_len[:, 0].max())
        out = self.fin_seq[:, 0, :L].clone()
        for b in range(self.B):
            out[b, int(self.fin_len[b, 0]):] = self.fill
        return out, self.fin_score[:, 0].clone()


def beam_search_generate(w, cfg: OracleConfig, inputs_embeds: Tensor, max_length: int, num_beams: int,
                         length_penalty: float = 1.0, early_stopping=True,
                         stop_ids: Optional[Sequence[int]] = None, mode: str = "fp32",
                         repetition_penalty: float = 1.0, return_scores: bool = False, do_sample: bool = False,
                         temperature: float = 1.0, top_p: float = 1.0, top_k: int = 0, min_length: int = 0):
    """generate(num_beams > 1): the prompt expanded to num_beams rows (repeat_interleave), BeamSearchState between the
    forward passes, the KV cache re-indexed by the surviving beams' parents.  Returns new tokens [B, L]
    (and the best scores [B] with return_scores)."""
    B, S0, _ = inputs_embeds.shape
    budget = max_length - S0
    if budget <= 0:
        raise ValueError("max_length must exceed the prompt length (HF raises here)")
    nb = int(num_beams)
    state = BeamSearchState(B, nb, cfg.vocab, budget, cfg.eos_token_id, cfg.pad_token_id, length_penalty,
                            early_stopping, stop_ids, repetition_penalty, do_sample, temperature, top_p, top_k,
                            min_new_tokens=max(int(min_length) - S0, 0))      # HF subtracts the prompt length first
    logits, cache = decoder_prefill(w, cfg, inputs_embeds.repeat_interleave(nb, dim=0), mode)
    while True:
        go_on, flat_parent, tokens = state.step(logits)
        if not go_on:
            break
        cache = [(k.index_select(0, flat_parent), v.index_select(0, flat_parent)) for k, v in cache]
        logits, cache = decoder_decode_step(w, cfg, tokens, cache, mode)
    out, scores = state.result()
    return (out, scores) if return_scores else out


def warp_scores(scores: Tensor, temperature: float = 1.0, top_p: float = 1.0, top_k: int = 0,
                min_tokens_to_keep: int = 1) -> Tensor:
    """HF TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper (generation/logits_process.py), the order
    GenerationMixin._get_logits_processor builds them in for do_sample=True.  The reference never passes top_k, but its
    pinned transformers==4.49.0 (pyproject.toml:18) defaults GenerationConfig.top_k to 50, so its sampling is top-k 50
    then top-p (starvector_base.py:230-232).  Under beam search HF sets min_tokens_to_keep = 2 (one EOS id + 1).
    Returns the warped scores (filtered entries -inf)."""
    scores = scores.float()
    if temperature != 1.0:
        scores = scores / temperature
    if top_k and top_k > 0:
        k = min(max(int(top_k), min_tokens_to_keep), scores.shape[-1])
        kth = torch.topk(scores, k, dim=-1).values[..., -1:]
        scores = scores.masked_fill(scores < kth, float("-inf"))
    if top_p is not None and top_p < 1.0:
        sorted_logits, sorted_idx = torch.sort(scores, descending=False, dim=-1)
        cum = sorted_logits.softmax(dim=-1).cumsum(dim=-1)
        remove = cum <= (1.0 - top_p)
        remove[..., -min_tokens_to_keep:] = False
        remove = remove.scatter(-1, sorted_idx, remove)
        scores = scores.masked_fill(remove, float("-inf"))
    return scores


def top_p_filtered_probs(logits: Tensor, temperature: float, top_p: float, top_k: int = 0) -> Tensor:
    """The distribution torch.multinomial draws from on the do_sample=True, num_beams=1 path: softmax of the warped
    scores (starvector_base.py:230-232 defaults top_p 0.9 / temperature 1; top_k: see warp_scores)."""
    return warp_scores(logits, temperature, top_p, top_k, 1).softmax(dim=-1)


def generate_im2svg_tokens(w, cfg: OracleConfig, image: Tensor, prompt_ids: Tensor, max_length: int,
                           stop_ids=None, mode: str = "fp32") -> Tensor:
    """starvector_base.py:243-257 up to (not including) tokenizer.batch_decode:
    cat([prompt_ids, generated]) as int64 [B, P+N]."""
    emb = prepare_generation_inputs(w, cfg, image, prompt_ids, mode)
    new = greedy_generate(w, cfg, emb, max_length, stop_ids, mode)
    return torch.cat([prompt_ids, new], dim=1)
