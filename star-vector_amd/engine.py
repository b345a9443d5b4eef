"""Python handle on the C-ABI engine: tensors in, tensors out (PyTorch is only the allocator/stream)."""
from __future__ import annotations

import ctypes as C
import threading
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import SvConfig, SvSampling, check


@dataclass
class EngineConfig:
    """Shapes of the path; defaults = StarVector-1B (StarVectorConfig, starvector_arch.py:96-131)."""
    image_size: int = 224
    patch_size: int = 14
    vit_width: int = 1024
    vit_layers: int = 23
    vit_heads: int = 16
    adapter_norm: str = "layer_norm"
    hidden: int = 2048
    n_layer: int = 24
    n_head: int = 16
    n_inner: int = 8192
    vocab: int = 49156
    n_positions: int = 8192
    max_batch: int = 32
    max_seq_len: int = 2048
    ln_eps: float = 1e-5
    # StarVector-8B: SigLIP tower + StarCoder2 (RoPE, GQA); defaults describe v1
    arch: str = "v1"
    n_kv_head: int = 1
    rope_theta: float = 1e6
    vit_mlp: int = 4096
    vit_eps: float = 1e-6
    sliding_window: int = 0          # v2: keys visible to a query (4096 for bigcode/starcoder2-7b); 0 = all
    weight_dtype: str = "bf16"       # "fp8_e4m3": decoder weights + lm_head quantised at load (per-row scales), BASELINE config 5
    exclusive_device: object = False # False / True / "auto" (2: on until a fused launch gives up, then off for good and the call re-run: sv_config.exclusive_device).
                                     # True: this engine alone launches kernels on its GPU while decoding (one process per GPU): enables the
                                     # all-blocks-resident fused launches (include/starvector_hip.h sv_config.exclusive_device); same tokens

    @property
    def query_length(self) -> int:
        return (self.image_size // self.patch_size) ** 2 + (1 if self.arch == "v1" else 0)

    @staticmethod
    def starvector_8b(max_batch: int = 16, max_seq_len: int = 4096) -> "EngineConfig":
        """siglip_384 (google/siglip-large-patch16-384) + bigcode/starcoder2-7b shapes."""
        return EngineConfig(image_size=384, patch_size=16, vit_width=1024, vit_layers=24, vit_heads=16, hidden=4608,
                            n_layer=32, n_head=36, n_inner=18432, vocab=49152 + 5, n_positions=16384,
                            max_batch=max_batch, max_seq_len=max_seq_len, arch="v2", n_kv_head=4, rope_theta=1e6,
                            vit_mlp=4096, vit_eps=1e-6, sliding_window=4096)


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need(t: torch.Tensor, dtype, what: str) -> torch.Tensor:
    if not t.is_cuda:
        raise ValueError(f"{what} must be a CUDA(HIP) tensor; the engine has no CPU path")
    if t.dtype != dtype:
        raise ValueError(f"{what} must be {dtype}, got {t.dtype}")
    return t.contiguous()


def _exclusive_code(v) -> int:
    """sv_config.exclusive_device: 0 (shared GPU), 1 (this engine owns it), 2 ("auto": optimistic, falls back for good at the first give-up)."""
    if isinstance(v, str):
        if v.lower() in ("auto", "optimistic", "2"):
            return 2
        return 1 if v.lower() in ("1", "true", "yes") else 0
    if v is True or v is False or v is None:
        return int(bool(v))
    return 2 if int(v) == 2 else int(bool(v))


class HipEngine:
    """Owns one ``sv_engine`` (weights repacked into library memory, paged KV pool, workspaces)."""

    def __init__(self, cfg: EngineConfig, device: Optional[int] = None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.StarVectorHipError("no HIP device visible; the StarVector engine requires a GPU")
        self.cfg = cfg
        self.device = torch.cuda.current_device() if device is None else int(device)
        c = SvConfig(cfg.image_size, cfg.patch_size, cfg.vit_width, cfg.vit_layers, cfg.vit_heads,
                     _lib.SV_NORM_LAYER if cfg.adapter_norm == "layer_norm" else _lib.SV_NORM_BATCH,
                     cfg.hidden, cfg.n_layer, cfg.n_head, cfg.n_inner, cfg.vocab, cfg.n_positions,
                     cfg.max_batch, cfg.max_seq_len, cfg.ln_eps, self.device,
                     _lib.SV_ARCH_V2 if cfg.arch == "v2" else _lib.SV_ARCH_V1, cfg.n_kv_head, cfg.rope_theta,
                     cfg.vit_mlp, cfg.vit_eps if cfg.arch == "v2" else cfg.ln_eps,
                     int(cfg.sliding_window) if cfg.arch == "v2" else 0,
                     {"bf16": 0, "fp8_e4m3": 1}[cfg.weight_dtype], _exclusive_code(getattr(cfg, "exclusive_device", False)))
        h = C.c_void_p()
        check(self.lib.sv_create(C.byref(c), C.byref(h)), "sv_create")
        self._h = h
        self._dev = torch.device("cuda", self.device)
        # One multi-call sequence at a time per engine: a padded batch run as slots (cb_reset ... cb_reset), a classic
        # generate, a scoring forward.  The library serialises single calls with its own mutex; THIS lock keeps two host
        # threads from interleaving their call sequences (thread B's cb_reset would release thread A's slots).
        self.call_lock = threading.RLock()

    def close(self):
        if getattr(self, "_h", None):
            self.lib.sv_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights -------------------------------------------------------------------------------
    def load_weight(self, name: str, t: torch.Tensor) -> None:
        if t.dtype == torch.bfloat16:
            dt = _lib.SV_DTYPE_BF16
        elif t.dtype == torch.float32:
            dt = _lib.SV_DTYPE_F32
        else:
            t = t.to(torch.float32)
            dt = _lib.SV_DTYPE_F32
        t = t.to(self._dev).contiguous()
        shape = (C.c_int64 * max(t.dim(), 1))(*(t.shape if t.dim() else (1,)))
        check(self.lib.sv_load_weight(self._h, name.encode(), _ptr(t), dt, max(t.dim(), 1), shape, _stream()),
              f"sv_load_weight({name})")

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True) -> None:
        """Ingest a reference state_dict (keys as saved by the reference, train/util.py:71)."""
        skipped = []
        for k, v in sd.items():
            if k.endswith("num_batches_tracked") or k.endswith(".attn.bias") or k.endswith(".attn.masked_bias") \
                    or ".visual_encoder.head." in k or k.endswith("rotary_emb.inv_freq"):
                continue
            try:
                self.load_weight(k, v)
            except KeyError:
                skipped.append(k)
        if strict and skipped:
            raise KeyError(f"unexpected keys in state_dict: {skipped[:5]}{'...' if len(skipped) > 5 else ''}")
        check(self.lib.sv_weights_complete(self._h), "sv_weights_complete")

    def expected_weights(self) -> List[Tuple[str, int, bool, bool]]:
        """(reference state_dict key, element count, required, loaded) for every tensor this engine expects, sorted by name."""
        n = self.lib.sv_weight_count(self._h)
        if n < 0:
            check(n, "sv_weight_count")
        out = []
        name = C.create_string_buffer(512)
        numel, req, loaded = C.c_int64(0), C.c_int32(0), C.c_int32(0)
        for i in range(n):
            check(self.lib.sv_weight_info(self._h, i, name, 512, C.byref(numel), C.byref(req), C.byref(loaded)), "sv_weight_info")
            out.append((name.value.decode(), int(numel.value), bool(req.value), bool(loaded.value)))
        return out

    def load_random_weights(self, seed: int = 1234, std: float = 0.02) -> None:
        """Random-init weights of this architecture, drawn ON THE GPU one tensor at a time (throughput runs; there is no network for
        checkpoints): Linear / embedding / position tensors and biases N(0, std), LayerNorm / BatchNorm scales (and running_var) 1,
        their biases (and running_mean) 0.  Each tensor has its own generator seeded by (seed, crc32(name)): the values do not depend
        on the enumeration order or on what else is loaded."""
        import zlib
        for name, numel, required, _ in self.expected_weights():
            if not required:
                continue                                   # the tied lm_head: the engine packs wte for it
            norm = ".ln_" in name or "norm" in name        # ln_1 / ln_2 / ln_pre / ln_vision / ln_f, layer_norm*, *layernorm, norm
            if norm and (name.endswith(".weight") or name.endswith("running_var")):
                t = torch.ones(numel, dtype=torch.bfloat16, device=self._dev)
            elif norm and (name.endswith(".bias") or name.endswith("running_mean")):
                t = torch.zeros(numel, dtype=torch.bfloat16, device=self._dev)
            else:
                g = torch.Generator(device=self._dev).manual_seed((int(seed) << 32) ^ zlib.crc32(name.encode()))
                t = torch.empty(numel, dtype=torch.float32, device=self._dev).normal_(0.0, std, generator=g).to(torch.bfloat16)
            self.load_weight(name, t)
        check(self.lib.sv_weights_complete(self._h), "sv_weights_complete")

    # ---- forward entry points -----------------------------------------------------------------
    def encode_image(self, image: torch.Tensor) -> torch.Tensor:
        image = _need(image, torch.bfloat16, "image")
        B = image.shape[0]
        if image.shape[1:] != (3, self.cfg.image_size, self.cfg.image_size):
            raise ValueError(f"image must be [B,3,{self.cfg.image_size},{self.cfg.image_size}], got {tuple(image.shape)}")
        out = torch.empty(B, self.cfg.query_length, self.cfg.vit_width, dtype=torch.bfloat16, device=image.device)
        check(self.lib.sv_encode_image(self._h, _ptr(image), B, _ptr(out), _stream()), "sv_encode_image")
        return out

    def adapter(self, x: torch.Tensor) -> torch.Tensor:
        x = _need(x, torch.bfloat16, "adapter input")
        B = x.shape[0]
        if x.shape[1:] != (self.cfg.query_length, self.cfg.vit_width):
            raise ValueError(f"adapter input must be [B,{self.cfg.query_length},{self.cfg.vit_width}]")
        out = torch.empty(B, self.cfg.query_length, self.cfg.hidden, dtype=torch.bfloat16, device=x.device)
        check(self.lib.sv_adapter(self._h, _ptr(x), B, _ptr(out), _stream()), "sv_adapter")
        return out

    def embed_tokens(self, ids: torch.Tensor) -> torch.Tensor:
        ids = _need(ids, torch.int64, "input_ids")
        out = torch.empty(*ids.shape, self.cfg.hidden, dtype=torch.bfloat16, device=ids.device)
        check(self.lib.sv_embed_tokens(self._h, _ptr(ids), ids.numel(), _ptr(out), _stream()), "sv_embed_tokens")
        return out

    def prepare_inputs(self, enc: torch.Tensor, prompt_ids: torch.Tensor) -> torch.Tensor:
        """a1 (starvector_base.py:203-221) without the concatenation: adapter(enc) and the prompt's token embeddings are written straight
        into one [B, T + P, hidden] inputs_embeds buffer (the reference builds it with torch.cat; same values, no ATen kernel)."""
        enc = _need(enc, torch.bfloat16, "adapter input")
        ids = _need(prompt_ids, torch.int64, "input_ids")
        B, T = enc.shape[0], self.cfg.query_length
        if enc.shape[1:] != (T, self.cfg.vit_width) or ids.dim() != 2 or ids.shape[0] != B:
            raise ValueError(f"prepare_inputs: enc [B,{T},{self.cfg.vit_width}] and prompt_ids [B,P] expected")
        P = ids.shape[1]
        out = torch.empty(B, T + P, self.cfg.hidden, dtype=torch.bfloat16, device=enc.device)
        check(self.lib.sv_adapter_into(self._h, _ptr(enc), B, _ptr(out), T + P, _stream()), "sv_adapter_into")
        check(self.lib.sv_embed_tokens_into(self._h, _ptr(ids), B, P, _ptr(out), T + P, T, _stream()), "sv_embed_tokens_into")
        return out

    def prefill(self, inputs_embeds: torch.Tensor) -> torch.Tensor:
        x = _need(inputs_embeds, torch.bfloat16, "inputs_embeds")
        B, S0, D = x.shape
        if D != self.cfg.hidden:
            raise ValueError("inputs_embeds hidden size mismatch")
        logits = torch.empty(B, self.cfg.vocab, dtype=torch.float32, device=x.device)
        check(self.lib.sv_prefill(self._h, _ptr(x), B, S0, _ptr(logits), _stream()), "sv_prefill")
        return logits

    def forward_logits(self, inputs_embeds: torch.Tensor, num_logits_to_keep: int = 0) -> torch.Tensor:
        """Scoring forward: bf16 logits [B, n, vocab] of the last n = num_logits_to_keep positions (0 = all positions)."""
        x = _need(inputs_embeds, torch.bfloat16, "inputs_embeds")
        B, S, D = x.shape
        if D != self.cfg.hidden:
            raise ValueError("inputs_embeds hidden size mismatch")
        n = int(num_logits_to_keep) if num_logits_to_keep and num_logits_to_keep > 0 else S
        out = torch.empty(B, n, self.cfg.vocab, dtype=torch.bfloat16, device=x.device)
        check(self.lib.sv_forward_logits(self._h, _ptr(x), B, S, n, _ptr(out), _stream()), "sv_forward_logits")
        return out

    def decode_step(self, tokens: torch.Tensor) -> torch.Tensor:
        tokens = _need(tokens.to(torch.int32), torch.int32, "tokens")
        B = tokens.numel()
        logits = torch.empty(B, self.cfg.vocab, dtype=torch.float32, device=tokens.device)
        check(self.lib.sv_decode_step(self._h, _ptr(tokens), B, _ptr(logits), _stream()), "sv_decode_step")
        return logits

    def generate(self, inputs_embeds: torch.Tensor, max_length: int, do_sample: bool = False,
                 temperature: float = 1.0, top_p: float = 1.0, eos_token_id: int = 0, pad_token_id: int = 0,
                 stop_ids: Optional[Sequence[int]] = None, seed: int = 0, sync_every: int = 32,
                 repetition_penalty: float = 1.0, num_beams: int = 1, length_penalty: float = 1.0,
                 early_stopping=False, top_k: int = 0, on_tokens=None, min_new_tokens: int = 0) -> torch.Tensor:
        """HF ``generate`` semantics for inputs_embeds: returns ONLY the new tokens, int64 [B, N].
        ``num_beams`` > 1 runs HF's beam search on device (``early_stopping``: False, True or "never"); with
        ``do_sample`` it is HF's beam-sample.  ``top_k`` (0 = off) is HF's TopKLogitsWarper, applied before top-p.
        ``on_tokens(tokens [B, n] int64 cpu, first_col)``: streaming callback, called every ``sync_every`` steps with the
        columns that became final and once more at the end.  ``min_new_tokens``: HF's MinLengthLogitsProcessor after the
        prompt length has been subtracted from ``min_length`` (EOS cannot be chosen before that many new tokens)."""
        x = _need(inputs_embeds, torch.bfloat16, "inputs_embeds")
        B, S0, D = x.shape
        if D != self.cfg.hidden:
            raise ValueError("inputs_embeds hidden size mismatch")
        max_new = max_length - S0
        if max_new <= 0:
            raise ValueError(f"max_length ({max_length}) must exceed the prompt length ({S0})")
        stops = list(stop_ids) if stop_ids else []
        arr = (C.c_int32 * max(len(stops), 1))(*stops) if stops else None
        sp = SvSampling(int(bool(do_sample)), float(temperature), float(top_p), int(max_length), int(eos_token_id),
                        int(pad_token_id), len(stops), C.cast(arr, C.POINTER(C.c_int32)) if stops else None,
                        int(seed) & 0xFFFFFFFFFFFFFFFF, int(sync_every), float(repetition_penalty),
                        int(num_beams), float(length_penalty), _early_code(early_stopping), int(top_k or 0), None, None,
                        max(int(min_new_tokens or 0), 0))
        cb, cb_errors = None, []
        if on_tokens is not None:
            cb, cb_errors = token_callback(on_tokens)          # keep a reference alive for the duration of the call
            sp.on_tokens = C.cast(cb, C.c_void_p)
        out = torch.empty(B, max_new, dtype=torch.int64, device=x.device)
        n = C.c_int32(0)
        check(self.lib.sv_generate(self._h, _ptr(x), B, S0, C.byref(sp), _ptr(out), C.byref(n), _stream()), "sv_generate")
        if cb_errors:
            raise cb_errors[0]
        return out[:, : n.value]

    # ---- continuous batching: one request per row of the engine's batch (include/starvector_hip.h, sv_cb_*) ----------------
    def cb_admit(self, inputs_embeds: torch.Tensor, requests: Sequence[dict]) -> List[int]:
        """Prompt pass of len(requests) new requests (inputs_embeds [n, S0, D] bf16, equal prompt length) into free slots while
        the live slots keep their KV cache; samples their first token.  Each request: dict(max_new_tokens, do_sample=False,
        temperature=1, top_p=1, top_k=0, seed=0, eos_token_id=0, pad_token_id=0, stop_ids=None, repetition_penalty=1,
        min_new_tokens=0).  Returns the slot ids.  Raises StarVectorBusy when slots or KV pages are short."""
        x = _need(inputs_embeds, torch.bfloat16, "inputs_embeds")
        n, S0, D = x.shape
        if D != self.cfg.hidden or n != len(requests):
            raise ValueError("inputs_embeds / requests mismatch")
        arr = (_lib.SvCbRequest * n)()
        for i, r in enumerate(requests):
            stops = list(r.get("stop_ids") or [])
            if len(stops) > 16:
                raise ValueError("stop sequence longer than 16 ids")
            a = arr[i]
            a.do_sample = int(bool(r.get("do_sample", False))); a.temperature = float(r.get("temperature", 1.0))
            a.top_p = float(r.get("top_p", 1.0)); a.top_k = int(r.get("top_k", 0) or 0)
            a.seed = int(r.get("seed", 0)) & 0xFFFFFFFFFFFFFFFF; a.max_new_tokens = int(r["max_new_tokens"])
            a.eos_token_id = int(r.get("eos_token_id", 0)); a.pad_token_id = int(r.get("pad_token_id", 0))
            a.min_new_tokens = int(r.get("min_new_tokens", 0) or 0); a.repetition_penalty = float(r.get("repetition_penalty", 1.0) or 1.0)
            a.n_stop = len(stops)
            for k, t in enumerate(stops):
                a.stop_ids[k] = int(t)
        slots = (C.c_int32 * n)()
        check(self.lib.sv_cb_admit(self._h, _ptr(x), n, S0, arr, slots, _stream()), "sv_cb_admit")
        return list(slots)

    def cb_step(self, n_steps: int = 8) -> int:
        """n_steps decode steps for every live slot (hipGraph replay); returns how many slots are still generating."""
        live = C.c_int32(0)
        check(self.lib.sv_cb_step(self._h, int(n_steps), C.byref(live), _stream()), "sv_cb_step")
        return live.value

    def cb_poll(self):
        """(live flags, tokens emitted so far) per slot, two lists of max_batch ints."""
        n = self.cfg.max_batch
        lv, st = (C.c_int32 * n)(), (C.c_int32 * n)()
        check(self.lib.sv_cb_poll(self._h, lv, st, n), "sv_cb_poll")
        return list(lv), list(st)

    def cb_read(self, slot: int, first: int, count: int) -> torch.Tensor:
        buf = (C.c_int64 * max(count, 1))()
        check(self.lib.sv_cb_read(self._h, int(slot), int(first), int(count), buf), "sv_cb_read")
        return torch.tensor(list(buf)[:count], dtype=torch.int64)

    def cb_release(self, slot: int) -> None:
        check(self.lib.sv_cb_release(self._h, int(slot)), "sv_cb_release")

    def cb_reset(self) -> None:
        check(self.lib.sv_cb_reset(self._h), "sv_cb_reset")

    def beam_history(self):
        """(parent beams, tokens), each int32 [n_steps, batch * num_beams], of the last beam-search ``generate``."""
        n, rows = C.c_int32(0), C.c_int32(0)
        check(self.lib.sv_beam_history(self._h, None, None, 0, C.byref(n), C.byref(rows)), "sv_beam_history")
        par, tok = (C.c_int32 * (n.value * rows.value))(), (C.c_int32 * (n.value * rows.value))()
        check(self.lib.sv_beam_history(self._h, par, tok, n.value, C.byref(n), C.byref(rows)), "sv_beam_history")
        shape = (n.value, rows.value)
        return torch.tensor(list(par), dtype=torch.int32).view(shape), torch.tensor(list(tok), dtype=torch.int32).view(shape)

    def last_timing(self) -> Dict[str, float]:
        buf = (C.c_double * 4)()
        check(self.lib.sv_last_timing(self._h, buf), "sv_last_timing")
        return {"ttft_ms": buf[0], "decode_ms": buf[1], "decode_steps": buf[2], "graph": bool(buf[3]), "graph_steps": int(buf[3])}

    def step_plan(self) -> Dict[str, int]:
        """What the last generate() call's decode step actually was (sv_debug_step_plan): kernel nodes of its captured graph and which
        of the fused launches ran -- the engine's own decisions, not a re-derivation from the configuration."""
        buf = (C.c_int32 * 4)()
        check(self.lib.sv_debug_step_plan(self._h, buf), "sv_debug_step_plan")
        return {"graph_kernel_nodes": int(buf[0]), "rowln_cattn_fused": bool(buf[1]), "greedy_in_lm_head": bool(buf[2]), "mlp_fused": bool(buf[3])}

    def set_exp(self, mask: int) -> None:
        """Experiment bit mask (SV_EXP) of the live engine: in-process A/B runs (tools/ab_exp.py)."""
        check(self.lib.sv_debug_set_exp(self._h, int(mask)), "sv_debug_set_exp")

    def debug_attn_trace(self) -> torch.Tensor:
        """[rows * kv heads * splits, 16] int64 wall-clock stamps (100 MHz) of the decode attention of the middle layer of the last
        step (engine created with SV_ATTN_TRACE=1 in the environment; include/starvector_hip.h, sv_debug_attn_trace)."""
        cap = 8192
        buf = (C.c_int64 * (cap * 16))()
        n = self.lib.sv_debug_attn_trace(self._h, buf, cap)
        if n < 0:
            check(n, "sv_debug_attn_trace")
        return torch.tensor(list(buf[: n * 16]), dtype=torch.int64).view(n, 16)

    def debug_occupy_cus(self, blocks: int, lds_bytes: int = 144 * 1024, ms: int = 300):
        """Test tenant (include/starvector_hip_debug.h, sv_debug_occupy_cus): pin `lds_bytes` of LDS on `blocks` CUs for `ms` milliseconds
        from a stream of its own; returns at once."""
        check(self.lib.sv_debug_occupy_cus(self._h, blocks, lds_bytes, ms), "sv_debug_occupy_cus")

    def debug_xcc_map(self, blocks: int, heavy: bool = False):
        """[blocks] list of XCC_IDs: where the blocks of a 1-D launch of 8-wave blocks ran (heavy: with the decode attention's LDS footprint and
        10 us of residence, so that a grid above the CU count runs in rounds; include/starvector_hip.h, sv_debug_xcc_map)."""
        buf = (C.c_int32 * blocks)()
        check(self.lib.sv_debug_xcc_map(self._h, blocks, 1 if heavy else 0, buf), "sv_debug_xcc_map")
        return list(buf)

    def debug_mlp_trace(self) -> torch.Tensor:
        """[blocks, 8] int64 wall-clock stamps (100 MHz) of the last fused MLP launch of the middle layer (engine created with
        SV_MLP_TRACE=1 in the environment; include/starvector_hip.h, sv_debug_mlp_trace)."""
        buf = (C.c_int64 * (1024 * 8))()
        n = self.lib.sv_debug_mlp_trace(self._h, buf, 1024)
        if n < 0:
            check(n, "sv_debug_mlp_trace")
        return torch.tensor(list(buf[: n * 8]), dtype=torch.int64).view(n, 8)

    def debug_kv_load(self, layer: int, kv: torch.Tensor, lens: Optional[torch.Tensor] = None) -> None:
        """Test surface of the decode attention (include/starvector_hip.h, sv_debug_kv_load): kv [B, S, 2 * n_kv_head * head_dim]
        bf16 (k heads | v heads, K as cached) becomes tokens 0..S-1 of `layer`'s paged KV; every row's position is set to S, or to
        lens[b] (int32 [B], <= S) for a ragged batch."""
        kv = _need(kv, torch.bfloat16, "kv")
        B, S, _ = kv.shape
        if lens is not None:
            lens = _need(lens, torch.int32, "lens")
            if lens.numel() != B or int(lens.max()) > S or int(lens.min()) < 0:
                raise ValueError("lens must be int32 [B] with 0 <= lens[b] <= S")
        check(self.lib.sv_debug_kv_load(self._h, int(layer), _ptr(kv) if S > 0 else C.c_void_p(0), B, S, _ptr(lens), _stream()),
              "sv_debug_kv_load")
        torch.cuda.current_stream().synchronize()          # the caller may free `kv` right away

    def debug_attn_decode(self, layer: int, qkv: torch.Tensor, advance: bool = True) -> torch.Tensor:
        """One launch of the decode attention of `layer` for the new token whose c_attn output is qkv [B, QKV] float32 (before
        RoPE); returns [B, n_head * head_dim] bf16 and (advance) steps the positions, so repeated calls walk the sequence."""
        qkv = _need(qkv, torch.float32, "qkv")
        B = qkv.shape[0]
        out = torch.empty(B, self.cfg.hidden, dtype=torch.bfloat16, device=qkv.device)     # n_head * head_dim == hidden
        check(self.lib.sv_debug_attn_decode(self._h, int(layer), _ptr(qkv), B, _ptr(out), int(bool(advance)), _stream()),
              "sv_debug_attn_decode")
        return out

    def profile_decode_step(self, B: int, iters: int = 5) -> Dict[str, Dict[str, float]]:
        """HIP-event time per decode step by kernel class (eager launches of the graph's kernels)."""
        buf = (C.c_double * 10)()
        check(self.lib.sv_profile_decode_step(self._h, B, iters, buf, _stream()), "sv_profile_decode_step")
        names = ["skinny_gemm", "attn_decode", "row_update_ln"]
        res = {n: {"ms_per_step": buf[2 * i], "launches_per_step": buf[2 * i + 1]} for i, n in enumerate(names)}
        res["event_pair_overhead_ms"] = buf[6]
        res["skinny_chain_ms_per_step"] = buf[7]
        res["others_chain_ms_per_step"] = buf[8]
        return res

    TTFT_STAGES = ("encoder_gemm", "encoder_attention", "encoder_rows", "adapter_gemm", "adapter_norm_and_token_gather",
                   "prefill_gemm", "gemm_remainder_rows", "prefill_attention", "prefill_rows", "lm_head", "first_token_select")

    def profile_ttft(self, image: Optional[torch.Tensor], prompt_ids: torch.Tensor, iters: int = 3) -> Dict[str, float]:
        """Where the time to first token goes (include/starvector_hip_debug.h, sv_profile_ttft): ms per stage of ONE pass
        image -> encoder -> adapter -> prompt rows -> prompt pass -> lm_head -> first greedy token, HIP events in front of every
        launch group.  image: bf16 [B,3,S,S] or None (text2svg); prompt_ids: int64 [B, P]."""
        ids = _need(prompt_ids, torch.int64, "prompt_ids")
        B, P = ids.shape
        img = _need(image, torch.bfloat16, "image") if image is not None else None
        buf = (C.c_double * 24)()
        check(self.lib.sv_profile_ttft(self._h, _ptr(img), B, _ptr(ids), P, int(iters), buf, _stream()), "sv_profile_ttft")
        res = {n: buf[2 * i] for i, n in enumerate(self.TTFT_STAGES)}
        res["launches"] = {n: int(round(buf[2 * i + 1])) for i, n in enumerate(self.TTFT_STAGES)}
        res["event_pair_overhead_ms"] = buf[22]
        res["first_to_last_event_ms"] = buf[23]
        return res


def token_callback(on_tokens):
    """Wrap ``on_tokens(tokens [B, n] int64 cpu, first_col)`` as an `sv_token_callback`.  An exception cannot unwind through
    the C frames of `sv_generate` (ctypes would print and drop it): the first one is kept, later bursts are skipped, and
    the caller re-raises it when `sv_generate` has returned.  Returns (C callback, list that receives the exception)."""
    errors = []

    def _cb(_user, ptr, batch, first_col, n_cols):
        if errors:
            return
        try:
            flat = torch.tensor(ptr[: batch * n_cols], dtype=torch.int64).view(batch, n_cols)
            on_tokens(flat, int(first_col))
        except BaseException as ex:                        # noqa: BLE001 -- re-raised by the caller
            errors.append(ex)

    return _lib.TOKEN_CALLBACK(_cb), errors


def _early_code(early_stopping) -> int:
    """HF's early_stopping: False | True | "never"  ->  0 | 1 | 2."""
    if early_stopping is True or early_stopping is False or early_stopping is None:
        return int(bool(early_stopping))
    if early_stopping == "never":
        return 2
    raise ValueError("`early_stopping` must be a boolean or 'never'")   # HF's own check


class HipBeamScorer:
    """The device-side beam-search bookkeeping on its own (what HF's ``_beam_search`` does between two forward passes).
    ``step(logits)`` consumes the [batch * num_beams, vocab] fp32 logits of one step and returns
    (done, parent_rows, tokens, running_scores); ``finalize()`` returns (tokens [batch, L], scores [batch])."""

    def __init__(self, batch: int, num_beams: int, vocab: int, max_new: int, eos_token_id: int, pad_token_id: int,
                 length_penalty: float = 1.0, early_stopping=False, repetition_penalty: float = 1.0,
                 stop_ids: Optional[Sequence[int]] = None, do_sample: bool = False, temperature: float = 1.0,
                 top_p: float = 1.0, top_k: int = 0, seed: int = 0, min_new_tokens: int = 0):
        self.lib = _lib.load()
        stops = list(stop_ids) if stop_ids else []
        arr = (C.c_int32 * max(len(stops), 1))(*stops) if stops else None
        cfg = _lib.SvBeamConfig(int(batch), int(num_beams), int(vocab), int(max_new), int(eos_token_id),
                                int(pad_token_id), _early_code(early_stopping), float(length_penalty),
                                float(repetition_penalty), len(stops),
                                C.cast(arr, C.POINTER(C.c_int32)) if stops else None, int(bool(do_sample)),
                                float(temperature), float(top_p), int(top_k or 0), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                max(int(min_new_tokens or 0), 0))
        self._h = C.c_void_p()
        self.rows, self.batch, self.vocab, self.max_new = batch * num_beams, batch, vocab, max_new
        check(self.lib.sv_beam_create(C.byref(cfg), C.byref(self._h)), "sv_beam_create")

    def step(self, logits: torch.Tensor):
        x = _need(logits, torch.float32, "logits")
        if x.shape != (self.rows, self.vocab):
            raise ValueError(f"logits must be [{self.rows}, {self.vocab}]")
        done = C.c_int32(0)
        par, tok, sc = (C.c_int32 * self.rows)(), (C.c_int32 * self.rows)(), (C.c_float * self.rows)()
        check(self.lib.sv_beam_step(self._h, _ptr(x), x.stride(0), C.byref(done), par, tok, sc, _stream()), "sv_beam_step")
        return bool(done.value), torch.tensor(list(par)), torch.tensor(list(tok)), torch.tensor(list(sc))

    def finalize(self):
        toks = (C.c_int64 * (self.batch * self.max_new))()
        sc = (C.c_float * self.batch)()
        n = C.c_int32(0)
        check(self.lib.sv_beam_finalize(self._h, toks, C.byref(n), sc, _stream()), "sv_beam_finalize")
        t = torch.tensor(list(toks), dtype=torch.int64).view(self.batch, self.max_new)[:, : n.value].contiguous()
        return t, torch.tensor(list(sc))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.sv_beam_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def bench_linear(M: int, N: int, K: int, act: str = "none", residual: bool = False, iters: int = 10) -> float:
    """Average microseconds of one big-M MFMA GEMM launch on pseudo-random operands (HIP events)."""
    lib = _lib.load()
    us = C.c_double(0.0)
    check(lib.sv_bench_linear(M, N, K, _lib.ACT[act], int(residual), iters, C.byref(us), _stream()), "sv_bench_linear")
    return us.value


# ---- single operators (used by the parity tests; one per SURVEY.md section 8a row) ----------------
def op_layernorm(x, gamma, beta, eps=1e-5):
    lib = _lib.load()
    x = _need(x, torch.bfloat16, "x"); M, D = x.shape
    y = torch.empty_like(x)
    check(lib.sv_op_layernorm(_ptr(x), _ptr(_need(gamma, torch.bfloat16, "gamma")),
                              _ptr(_need(beta, torch.bfloat16, "beta")), _ptr(y), M, D, eps, _stream()))
    return y


def op_linear(x, W, bias=None, residual=None, act="none", out_f32=False):
    lib = _lib.load()
    x = _need(x, torch.bfloat16, "x"); W = _need(W, torch.bfloat16, "W")
    M, K = x.shape; N = W.shape[0]
    y = torch.empty(M, N, dtype=torch.float32 if out_f32 else torch.bfloat16, device=x.device)
    b = _need(bias, torch.bfloat16, "bias") if bias is not None else None
    r = _need(residual, torch.bfloat16, "residual") if residual is not None else None
    check(lib.sv_op_linear(_ptr(x), _ptr(W), _ptr(b), _ptr(r), _ptr(y), M, N, K, _lib.ACT[act], int(out_f32), _stream()))
    return y


def set_gemm_form(form: int) -> None:
    """-1: the tuned choice (default); 0 / 1: every big-M GEMM launch of the process takes 128^2 tiles / 256^2 tiles, rows not peeled;
    2: 256^2 tiles + the row remainder through the tail kernel (sv_debug_set_gemm_form).
    Same bits every way; test and A/B surface."""
    check(_lib.load().sv_debug_set_gemm_form(int(form)))


def set_linear_seq_rows(seq_rows: int) -> None:
    """Sequence structure `linear` / `bench_linear` hand to the big-M dispatch (sv_debug_set_linear_seq_rows): S > 0 = the rows are
    sequences of S rows (where the per-sequence form holds for the projection, the rows a sequence leaves over its 256-row tiles take
    the split-K remainder kernel), S < 0 = the rows are the last rows of sequences of |S| rows, 0 = none (default)."""
    check(_lib.load().sv_debug_set_linear_seq_rows(int(seq_rows)))


def gemm_seq_form(S: int, N: int, K: int, act: str = "none") -> bool:
    """Does the projection (N, K, act) take the per-sequence remainder form for sequences of S rows (sv_debug_gemm_seq_form; host arithmetic)?"""
    rc = _lib.load().sv_debug_gemm_seq_form(int(S), int(N), int(K), _lib.ACT[act])
    if rc < 0:
        check(rc, "sv_debug_gemm_seq_form")
    return rc == 1


def set_skinny_form(form: int) -> None:
    """Kernel of the 33..64-row decode GEMMs (sv_debug_set_skinny_form): 0 registers only, 1 LDS ring (default), 2 / 3 its fixed depths."""
    check(_lib.load().sv_debug_set_skinny_form(int(form)))


def set_op_col_tiles(col_tiles: int) -> None:
    """Column tiles per block (1..3, 0 = default) of the op-level decode GEMMs at 33..64 rows (sv_debug_set_col_tiles)."""
    check(_lib.load().sv_debug_set_col_tiles(int(col_tiles)))


def op_linear_skinny(x, W, bias=None, splitk=1):
    lib = _lib.load()
    x = _need(x, torch.bfloat16, "x"); W = _need(W, torch.bfloat16, "W")
    M, K = x.shape; N = W.shape[0]
    y = torch.empty(M, N, dtype=torch.float32, device=x.device)
    b = _need(bias, torch.bfloat16, "bias") if bias is not None else None
    check(lib.sv_op_linear_skinny(_ptr(x), _ptr(W), _ptr(b), _ptr(y), M, N, K, splitk, _stream()))
    return y


def op_linear_skinny_fp8(x, W, bias=None, splitk=1):
    """Returns (y fp32 [M, N], row scales fp32 [N]) of the fp8-weight decode GEMM (weights quantised on device)."""
    lib = _lib.load()
    x = _need(x, torch.bfloat16, "x"); W = _need(W, torch.bfloat16, "W")
    M, K = x.shape; N = W.shape[0]
    y = torch.empty(M, N, dtype=torch.float32, device=x.device)
    sc = torch.empty(N, dtype=torch.float32, device=x.device)
    b = _need(bias, torch.bfloat16, "bias") if bias is not None else None
    check(lib.sv_op_linear_skinny_fp8(_ptr(x), _ptr(W), _ptr(b), _ptr(y), _ptr(sc), M, N, K, splitk, _stream()))
    return y, sc


def op_linear_skinny_epi(x, W, bias=None, act="none", out_f32=False):
    """The decode GEMM's fused epilogues (split-K 1): act(bf16(x W^T + bias)) as bf16 rows (the c_fc form), or with out_f32
    the lm_head form: float32 rows of x W^T holding bf16-rounded values (no bias)."""
    lib = _lib.load()
    x = _need(x, torch.bfloat16, "x"); W = _need(W, torch.bfloat16, "W")
    M, K = x.shape; N = W.shape[0]
    y = torch.empty(M, N, dtype=torch.float32 if out_f32 else torch.bfloat16, device=x.device)
    b = _need(bias, torch.bfloat16, "bias") if bias is not None else None
    check(lib.sv_op_linear_skinny_epi(_ptr(x), _ptr(W), _ptr(b), _ptr(y), M, N, K, _lib.ACT[act], int(out_f32), _stream()),
          "sv_op_linear_skinny_epi")
    return y


def op_lm_head_argmax(x, W):
    """The lm_head form of the decode GEMM with the greedy selection folded into its epilogue (what a plain greedy generate
    runs per step): returns (logits float32 [M, N] holding bf16-rounded values, arg-max column per row as an int64 cpu tensor)."""
    lib = _lib.load()
    x = _need(x, torch.bfloat16, "x"); W = _need(W, torch.bfloat16, "W")
    M, K = x.shape; N = W.shape[0]
    y = torch.empty(M, N, dtype=torch.float32, device=x.device)
    idx = (C.c_int32 * M)()
    check(lib.sv_op_lm_head_argmax(_ptr(x), _ptr(W), _ptr(y), idx, M, N, K, _stream()), "sv_op_lm_head_argmax")
    return y, torch.tensor(list(idx), dtype=torch.int64)


def op_decode_proj_fold(x, Wp, bp, h, gamma, beta, Wf, bf_, act="gelu_tanh", eps=1e-5):
    """The 6-launch decode layer's two kernels (csrc/decode_cols.hip): returns (h2 bf16 [M, D], y bf16 [M, F])."""
    lib = _lib.load()
    t = lambda v, n: _need(v, torch.bfloat16, n)
    x, Wp, h, gamma, beta, Wf = t(x, "x"), t(Wp, "Wp"), t(h, "h"), t(gamma, "gamma"), t(beta, "beta"), t(Wf, "Wf")
    bp = t(bp, "bp") if bp is not None else None
    bf_ = t(bf_, "bf") if bf_ is not None else None
    M, Kp = x.shape; D = Wp.shape[0]; F = Wf.shape[0]
    h2 = torch.empty(M, D, dtype=torch.bfloat16, device=x.device)
    y = torch.empty(M, F, dtype=torch.bfloat16, device=x.device)
    check(lib.sv_op_decode_proj_fold(_ptr(x), _ptr(Wp), _ptr(bp), _ptr(h), _ptr(gamma), _ptr(beta), float(eps), _ptr(Wf), _ptr(bf_),
                                     _ptr(h2), _ptr(y), M, D, Kp, F, _lib.ACT[act], _stream()), "sv_op_decode_proj_fold")
    return h2, y


def bench_decode_linear(M: int, N: int, K: int, splitk: int = 1, mode: int = 0, iters: int = 100) -> float:
    """Average microseconds per launch of one decode GEMM (mode 0 fp32 slabs, 1 bias + GELU, 2 fp32 logits)."""
    lib = _lib.load()
    us = C.c_double(0.0)
    check(lib.sv_bench_decode_linear(M, N, K, splitk, mode, iters, C.byref(us), _stream()), "sv_bench_decode_linear")
    return us.value


def op_cvt_bf16_hw(x):
    lib = _lib.load()
    x = _need(x, torch.float32, "x")
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    check(lib.sv_op_cvt_bf16_hw(_ptr(x), _ptr(y), x.numel(), _stream()))
    return y


def op_attention(q, k, v, n_head, n_kv_head, causal, scale=None):
    lib = _lib.load()
    q = _need(q, torch.bfloat16, "q"); k = _need(k, torch.bfloat16, "k"); v = _need(v, torch.bfloat16, "v")
    B, S, HD = q.shape
    hd = HD // n_head
    out = torch.empty_like(q)
    check(lib.sv_op_attention(_ptr(q), _ptr(k), _ptr(v), _ptr(out), B, S, n_head, n_kv_head, hd, int(causal),
                              float(scale if scale is not None else hd ** -0.5), _stream()))
    return out


def op_plane_layernorm(x, gamma, beta, eps=1e-5):
    lib = _lib.load()
    x = _need(x, torch.bfloat16, "x"); B = x.shape[0]; QD = x[0].numel()
    y = torch.empty_like(x)
    check(lib.sv_op_plane_layernorm(_ptr(x), _ptr(_need(gamma, torch.bfloat16, "gamma")),
                                    _ptr(_need(beta, torch.bfloat16, "beta")), _ptr(y), B, QD, eps, _stream()))
    return y


def op_argmax(logits):
    lib = _lib.load()
    logits = _need(logits, torch.float32, "logits"); B, V = logits.shape
    if V % 4:
        raise ValueError("V must be a multiple of 4")
    out = torch.empty(B, dtype=torch.int32, device=logits.device)
    check(lib.sv_op_argmax(_ptr(logits), B, V, V, _ptr(out), _stream()))
    return out


def op_preprocess_image(pixels: torch.Tensor, size: int, mean, std, recipe: str = "starvector") -> torch.Tensor:
    """uint8 [H, W, 3|4] on the GPU -> float32 [3, size, size].  recipe "starvector": composite on white, pad to square,
    Pillow-exact bicubic resize, ToTensor, Normalize (starvector/data/util.py:40-68); "siglip": HF SiglipImageProcessor
    (alpha dropped, stretch-resize, rescale 1/255, normalise) as the v2 tower uses it (image_encoder.py:45-48)."""
    lib = _lib.load()
    px = _need(pixels, torch.uint8, "pixels")
    if px.dim() != 3 or px.shape[2] not in (3, 4):
        raise ValueError("pixels must be uint8 [H, W, 3] (RGB) or [H, W, 4] (RGBA)")
    out = torch.empty(3, size, size, dtype=torch.float32, device=px.device)
    m3, s3 = (C.c_float * 3)(*[float(v) for v in mean]), (C.c_float * 3)(*[float(v) for v in std])
    check(lib.sv_preprocess_image(_ptr(px), px.shape[1], px.shape[0], px.shape[2], int(size),
                                  {"starvector": 0, "siglip": 1}[recipe], m3, s3, _ptr(out), _stream()), "sv_preprocess_image")
    return out


def op_preprocess_images(pixels, size: int, mean, std, recipe: str = "starvector") -> torch.Tensor:
    """A batch of uint8 [H, W, 3|4] GPU tensors (any sizes) -> float32 [n, 3, size, size], one call (`sv_preprocess_images`):
    three launches per 32 images, no host copy, no synchronisation, workspace from torch's caching allocator."""
    lib = _lib.load()
    if len(pixels) == 0:
        raise ValueError("empty batch")
    px = [_need(t, torch.uint8, "pixels") for t in pixels]
    for t in px:
        if t.dim() != 3 or t.shape[2] not in (3, 4):
            raise ValueError("pixels must be uint8 [H, W, 3] (RGB) or [H, W, 4] (RGBA)")
    n = len(px)
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in px])
    ws_ = (C.c_int32 * n)(*[t.shape[1] for t in px])
    hs_ = (C.c_int32 * n)(*[t.shape[0] for t in px])
    cs_ = (C.c_int32 * n)(*[t.shape[2] for t in px])
    rc = {"starvector": 0, "siglip": 1}[recipe]
    need = lib.sv_preprocess_workspace_bytes(ws_, hs_, n, int(size), rc)
    if need < 0:
        check(int(need), "sv_preprocess_workspace_bytes")
    work = torch.empty(int(need), dtype=torch.uint8, device=px[0].device)
    out = torch.empty(n, 3, size, size, dtype=torch.float32, device=px[0].device)
    m3, s3 = (C.c_float * 3)(*[float(v) for v in mean]), (C.c_float * 3)(*[float(v) for v in std])
    check(lib.sv_preprocess_images(ptrs, ws_, hs_, cs_, n, int(size), rc, m3, s3, _ptr(out), _ptr(work), int(need), _stream()),
          "sv_preprocess_images")
    return out


def op_sample_top_p(logits, temperature, top_p, seed, step, top_k=0):
    lib = _lib.load()
    logits = _need(logits, torch.float32, "logits"); B, V = logits.shape
    out = torch.empty(B, dtype=torch.int32, device=logits.device)
    check(lib.sv_op_sample(_ptr(logits), B, V, V, float(temperature), int(top_k), float(top_p), int(seed), int(step),
                           _ptr(out), _stream()))
    return out
