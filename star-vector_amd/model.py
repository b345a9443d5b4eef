"""Host-side mirror of the reference's operator interface for the im2svg hot path.

Same names, argument meaning and error behaviour as the reference classes, so a caller of
``StarVectorForCausalLM.generate_im2svg`` / ``model.model.image_encoder`` / ``image_projection`` /
``svg_transformer.transformer.generate`` can switch without edits.  Every forward goes through the
C-ABI HIP engine (``engine.HipEngine``); there is no PyTorch compute path here.

Reference lines mirrored:
  StarVectorConfig / StarVectorForCausalLM   starvector/model/starvector_arch.py:96-193
  StarVectorBase orchestration               starvector/model/models/starvector_base.py:203-295
  StarVectorStarCoder (v1)                   starvector/model/models/starvector_v1.py:6-22
  ImageEncoder (clip branch)                 starvector/model/image_encoder/image_encoder.py:9-119
  Adapter                                    starvector/model/adapters/adapter.py:12-39
  StarCoderModel                             starvector/model/llm/starcoder.py:9-53
  ImageTrainProcessor                        starvector/data/util.py:40-68
"""
from __future__ import annotations

import json
import os
import contextlib
import threading
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from .engine import EngineConfig, HipEngine

# Set while the calling thread runs an exclusive job of a ContinuousBatcher (beam search, multi-row batches): inside it the
# mirror talks to the engine directly; every OTHER thread keeps going through the batcher's queue.
_EXCLUSIVE = threading.local()


def _in_exclusive_job() -> bool:
    return bool(getattr(_EXCLUSIVE, "active", False))

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)   # data/util.py:33-38
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class StarVectorConfig:
    """starvector_arch.py:96-131 (fields the hot path reads) + the engine sizing knobs."""
    model_type = "starvector"

    def __init__(self, starcoder_model_name: str = "bigcode/starcoderbase-1b", image_encoder_type: str = "clip",
                 adapter_norm: str = "layer_norm", image_size: int = 224, max_length: int = 8192,
                 max_length_train: int = 8192, use_flash_attn: bool = True, use_cache: bool = True,
                 num_attention_heads: int = 16, num_hidden_layers: int = 24, vocab_size: int = 49152,
                 hidden_size: int = 2048, num_kv_heads: int = 4, torch_dtype: str = "bfloat16",
                 # engine-only (not in the reference config): shapes that the reference takes from the HF
                 # sub-model configs, and the batch/sequence capacity the KV pool is sized for
                 n_inner: Optional[int] = None, n_positions: Optional[int] = None, added_tokens: Optional[int] = None,
                 vit_width: int = 1024, vit_layers: int = 23, vit_heads: int = 16, patch_size: int = 14,
                 max_batch: int = 32, exclusive_device: Optional[bool] = None, **kwargs):
        self.starcoder_model_name = starcoder_model_name
        self.image_encoder_type = image_encoder_type
        self.adapter_norm = adapter_norm
        self.image_size = image_size
        self.max_length = max_length
        self.max_length_train = max_length_train
        self.use_flash_attn = use_flash_attn
        self.use_cache = use_cache
        self.num_attention_heads = num_attention_heads
        self.num_hidden_layers = num_hidden_layers
        self.vocab_size = vocab_size
        self.hidden_size = hidden_size
        self.num_kv_heads = num_kv_heads
        self.torch_dtype = torch_dtype
        self.n_inner = n_inner if n_inner is not None else 4 * hidden_size
        v2 = "starcoder2" in starcoder_model_name
        # bigcode/starcoderbase-1b: 8192 learned positions; bigcode/starcoder2-7b: max_position_embeddings 16384
        self.n_positions = n_positions if n_positions is not None else (16384 if v2 else 8192)
        # [PAD] + <svg-start>,<image-start>,<caption-start> (llm/starcoder.py:40-53); v2 adds <svg-end> (llm/starcoder2.py:47)
        self.added_tokens = added_tokens if added_tokens is not None else (5 if v2 else 4)
        self.vit_width, self.vit_layers, self.vit_heads, self.patch_size = vit_width, vit_layers, vit_heads, patch_size
        self.max_batch = max_batch
        # one process per GPU and nothing else decoding on it (the deployment the engine is built for): allows the launches whose blocks
        # wait for each other (INTEGRATION.md "Deployment knob").  None = the environment decides (SV_EXCLUSIVE_DEVICE=1), default off
        # None: SV_EXCLUSIVE_DEVICE (0 / 1 / auto), default "auto" = the fused decode launches on until one of them finds the GPU shared, then off for good
        # with the failed call re-run (include/starvector_hip.h, sv_config.exclusive_device = 2): same tokens either way
        env = os.environ.get("SV_EXCLUSIVE_DEVICE", "auto")
        self.exclusive_device = ({"0": False, "1": True}.get(env, "auto")) if exclusive_device is None else exclusive_device
        for k, v in kwargs.items():
            setattr(self, k, v)

    @staticmethod
    def _visible_keys(window: int, semantics: str) -> int:
        if semantics not in ("flash_attention_2", "sdpa", "eager"):
            raise ValueError(f"window_semantics={semantics!r}: 'flash_attention_2' (W + 1 keys, transformers 4.49's FA2 call, "
                             "what the reference loads) or 'sdpa' / 'eager' (W keys)")
        window = int(window or 0)
        return window + 1 if (window > 0 and semantics == "flash_attention_2") else window

    @property
    def is_v2(self) -> bool:                      # starvector_arch.py:139-144 picks the class the same way
        return "starcoder2" in self.starcoder_model_name

    def engine_config(self) -> EngineConfig:
        dt = str(self.torch_dtype).replace("torch.", "")
        if dt in ("float16", "half", "float32", "float"):
            # the reference's 1B validation config is fp16 (configs/generation/hf/starvector-1b/im2svg.yaml:16) and
            # scripts/quickstart.py:16 casts the image to fp16: such callers keep working -- weights and inputs are converted at
            # the boundary, the engine computes in bfloat16 (same exponent range as fp32; token streams may differ from an fp16
            # run at near-ties, which is what any dtype change does)
            import warnings
            warnings.warn(f"torch_dtype={self.torch_dtype}: the HIP engine computes in bfloat16; weights and inputs are "
                          "converted at the boundary", stacklevel=2)
        elif dt not in ("bfloat16", "auto"):
            raise ValueError(f"unsupported torch_dtype={self.torch_dtype} (bfloat16, or float16 / float32 converted to it)")
        if self.is_v2:
            # StarVector-8B: siglip_384 tower + StarCoder2 decoder (configs/models/starvector-8b/im2svg-stack.yaml)
            if self.image_encoder_type != "siglip_384":
                raise NotImplementedError(f"image_encoder_type={self.image_encoder_type!r}: v2 is built for siglip_384")
            g = lambda k, d: getattr(self, k, d)
            return EngineConfig(image_size=g("siglip_image_size", 384), patch_size=g("siglip_patch_size", 16),
                                vit_width=self.vit_width, vit_layers=g("siglip_layers", 24), vit_heads=self.vit_heads,
                                adapter_norm=self.adapter_norm, hidden=self.hidden_size, n_layer=self.num_hidden_layers,
                                n_head=self.num_attention_heads, n_inner=self.n_inner,
                                vocab=self.vocab_size + self.added_tokens, n_positions=self.n_positions,
                                max_batch=self.max_batch, max_seq_len=min(self.max_length, self.n_positions), arch="v2",
                                n_kv_head=self.num_kv_heads, rope_theta=g("rope_theta", 1e6),
                                vit_mlp=g("siglip_mlp", 4096), vit_eps=1e-6,
                                # bigcode/starcoder2-7b config.json: sliding_window 4096.  The reference loads the decoder with
                                # attn_implementation="flash_attention_2" (llm/starcoder2.py:21-26); transformers 4.49 hands FA2
                                # window_size=(W, W), i.e. the query sees W + 1 keys, while its eager/sdpa mask (and later releases
                                # everywhere) show W.  `window_semantics` picks which one the engine's window (= number of visible
                                # keys) mirrors: "flash_attention_2" (default: what the reference runs) or "sdpa".
                                sliding_window=self._visible_keys(g("sliding_window", 4096), g("window_semantics", "flash_attention_2")),
                                exclusive_device=self.exclusive_device)
        if self.image_encoder_type != "clip":
            raise NotImplementedError(f"image_encoder_type={self.image_encoder_type!r}: v1 is built for the clip branch")
        return EngineConfig(image_size=self.image_size, patch_size=self.patch_size, vit_width=self.vit_width,
                            vit_layers=self.vit_layers, vit_heads=self.vit_heads, adapter_norm=self.adapter_norm,
                            hidden=self.hidden_size, n_layer=self.num_hidden_layers, n_head=self.num_attention_heads,
                            n_inner=self.n_inner, vocab=self.vocab_size + self.added_tokens,
                            n_positions=self.n_positions, max_batch=self.max_batch,
                            max_seq_len=min(self.max_length, self.n_positions), exclusive_device=self.exclusive_device)


def config_from_checkpoint(cfg_json: Dict, shapes: Dict[str, tuple]) -> StarVectorConfig:
    """`config.json` of a reference checkpoint + the shapes of its tensors -> StarVectorConfig.  The reference takes the
    decoder's vocabulary (after `resize_token_embeddings`), position count and MLP width from the HF sub-model it
    instantiates (llm/starcoder.py:33-53, llm/starcoder2.py:22-53); offline they are read off the saved tensors, which
    is what those numbers ended up as.  Explicit entries of `cfg_json` win."""
    cfg = StarVectorConfig(**{**cfg_json, "torch_dtype": "bfloat16"})
    dec = "model.svg_transformer.transformer." + ("model." if cfg.is_v2 else "transformer.")
    emb = shapes.get(dec + ("embed_tokens.weight" if cfg.is_v2 else "wte.weight"))
    if emb is not None:
        if "hidden_size" not in cfg_json:
            cfg.hidden_size = int(emb[1])
        if "added_tokens" not in cfg_json:
            cfg.added_tokens = int(emb[0]) - int(cfg.vocab_size)
            if cfg.added_tokens < 0:
                raise ValueError(f"checkpoint embeds {emb[0]} tokens, fewer than vocab_size {cfg.vocab_size}")
    wpe = shapes.get(dec + "wpe.weight")
    if wpe is not None and "n_positions" not in cfg_json:
        cfg.n_positions = int(wpe[0])
    fc = shapes.get(dec + ("layers.0." if cfg.is_v2 else "h.0.") + "mlp.c_fc.weight")
    if fc is not None and "n_inner" not in cfg_json:
        cfg.n_inner = int(fc[0])
    elif "n_inner" not in cfg_json:
        cfg.n_inner = 4 * cfg.hidden_size
    return cfg


# --------------------------------------------------------------------------------------------------
# tokenizer used when the gated StarCoder tokenizer is not on disk
# --------------------------------------------------------------------------------------------------
class _Encoding(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def to(self, device):
        return _Encoding({k: (v.to(device) if torch.is_tensor(v) else v) for k, v in self.items()})


class ByteTokenizer:
    """Deterministic byte-level stand-in with the tokenizer surface the path uses
    (llm/starcoder.py:40-53): eos id 0, ``[PAD]`` = vocab_size, then the three added tokens."""

    def __init__(self, vocab_size: int = 49152, v2: bool = False):
        self.vocab_size = vocab_size
        self.eos_token, self.eos_token_id = "<|endoftext|>", 0
        self.bos_token_id = 0
        self.pad_token, self.pad_token_id = "[PAD]", vocab_size
        self.added = {"<svg-start>": vocab_size + 1, "<image-start>": vocab_size + 2, "<caption-start>": vocab_size + 3}
        if v2:                                    # llm/starcoder2.py:47 adds <svg-end> too, pads on the left (:53)
            self.added["<svg-end>"] = vocab_size + 4
        self.padding_side = "left" if v2 else "right"

    def __len__(self):
        return self.vocab_size + 1 + len(self.added)

    def encode(self, text: str, add_special_tokens: bool = False) -> List[int]:
        if text in self.added:
            return [self.added[text]]
        return [1 + b for b in text.encode("utf-8")]

    def __call__(self, text, add_special_tokens: bool = False, return_tensors: Optional[str] = None, **kw):
        single = isinstance(text, str)
        rows = [self.encode(t) for t in ([text] if single else text)]
        if return_tensors is None:
            return _Encoding(input_ids=rows[0] if single else rows)
        n = max(len(r) for r in rows)
        ids = torch.full((len(rows), n), self.pad_token_id, dtype=torch.long)
        mask = torch.zeros((len(rows), n), dtype=torch.long)
        for i, r in enumerate(rows):
            if getattr(self, "padding_side", "right") == "left":      # llm/starcoder2.py:53
                ids[i, n - len(r):] = torch.tensor(r, dtype=torch.long)
                mask[i, n - len(r):] = 1
            else:
                ids[i, : len(r)] = torch.tensor(r, dtype=torch.long)
                mask[i, : len(r)] = 1
        return _Encoding(input_ids=ids, attention_mask=mask)

    def decode(self, ids, skip_special_tokens: bool = True) -> str:
        out = bytearray()
        for t in (ids.tolist() if torch.is_tensor(ids) else ids):
            if 1 <= t <= 256:
                out.append(t - 1)
            elif not skip_special_tokens:
                out.extend(f"<{t}>".encode())
        return out.decode("utf-8", "replace")

    def batch_decode(self, ids, skip_special_tokens: bool = True) -> List[str]:
        return [self.decode(r, skip_special_tokens) for r in ids]


# --------------------------------------------------------------------------------------------------
# image pre-processing (host side, PIL): data/util.py:40-68
# --------------------------------------------------------------------------------------------------
class ImageTrainProcessor:
    """starvector/data/util.py:40-68.  ``device=None`` (default): the reference's own host recipe with Pillow.
    ``device="cuda"``: the same arithmetic on the GPU (``sv_preprocess_image``: composite, pad, Pillow-exact bicubic
    resize, /255, normalise) -- the float32 tensor is bit-identical, it just never leaves the device."""

    def __init__(self, mean=None, std=None, size: int = 224, device=None, **kwargs):
        self._mean, self._std = tuple(mean or CLIP_MEAN), tuple(std or CLIP_STD)
        self.mean = torch.tensor(self._mean).view(3, 1, 1)
        self.std = torch.tensor(self._std).view(3, 1, 1)
        self.size = size
        self.device = device

    def _pixels(self, img):
        import numpy as np
        # engine-side processor: always the device kernels (palette / grey / CMYK images are only re-encoded as RGB(A)
        # first -- the reference itself cannot normalise a non-RGB image with its 3-channel mean)
        if img.mode not in ("RGB", "RGBA"):
            img = img.convert("RGBA" if "transparency" in img.info or img.mode in ("LA", "PA") else "RGB")
        return torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).to(self.device, non_blocking=True)

    def batch(self, images) -> torch.Tensor:
        """A list of PIL images -> float32 [n, 3, size, size] in ONE device call (`sv_preprocess_images`): the serving path
        pre-processes a whole request batch with three launches instead of one synchronised round trip per image."""
        if self.device is None:
            return torch.stack([self(img) for img in images], 0)
        from .engine import op_preprocess_images
        return op_preprocess_images([self._pixels(img) for img in images], self.size, self._mean, self._std)

    def __call__(self, img):
        from PIL import Image
        import numpy as np
        if self.device is not None:
            from .engine import op_preprocess_image
            return op_preprocess_image(self._pixels(img), self.size, self._mean, self._std)
        if img.mode == "RGBA":                               # _rgba_to_rgb_white (data/util.py:64-67)
            bg = Image.new("RGB", img.size, (255, 255, 255))
            bg.paste(img, mask=img.split()[3])
            img = bg
        w, h = img.size                                      # _pad_to_square, white (data/util.py:56-62)
        m = max(w, h)
        if (w, h) != (m, m):
            canvas = Image.new(img.mode, (m, m), 255 if img.mode in ("L", "1") else (255,) * len(img.getbands()))
            canvas.paste(img, ((m - w) // 2, (m - h) // 2))
            img = canvas
        if img.size != (self.size, self.size):               # transforms.Resize(size, BICUBIC) on a PIL image
            img = img.resize((self.size, self.size), Image.BICUBIC)
        if img.mode != "RGB":
            img = img.convert("RGB")
        x = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float() / 255.0   # ToTensor
        x = (x - self.mean) / self.std
        return x if self.device is None else x.to(self.device)


class SimpleStarVectorProcessor:
    """starvector_arch.py:17-90: the HF-style processor a v1 checkpoint's `AutoProcessor` resolves to, reached as
    `starvector.model.processor` (starvector_v1.py:10; scripts/quickstart-hf.py and the validation dataset call it as
    `processor(image, return_tensors="pt")["pixel_values"]`).  Same pixels as ImageTrainProcessor except for RGBA input:
    here the alpha band is DROPPED (`img.convert("RGB")`, :41), not composited on white.  One image -> [3, S, S], a list ->
    [B, 3, S, S]; `text` goes through the tokenizer with the reference's arguments (:76-83).
    With `device` set the resize / normalise run on that GPU (`sv_preprocess_image`), bit-identical to the host recipe."""

    def __init__(self, tokenizer=None, size: int = 224, mean=None, std=None, device=None, **kwargs):
        self.tokenizer = tokenizer
        self.mean, self.std, self.size = tuple(mean or CLIP_MEAN), tuple(std or CLIP_STD), size
        self._pixels = ImageTrainProcessor(mean=self.mean, std=self.std, size=size, device=device)

    def _one(self, img):
        if img.mode == "RGBA":
            img = img.convert("RGB")
        return self._pixels(img)

    def __call__(self, images=None, text=None, max_length=None, **kwargs):
        if images is None and text is None:
            raise ValueError("You have to specify at least one of `images` or `text`.")
        data = {}
        if text is not None:
            data.update(self.tokenizer(text, truncation=True, add_special_tokens=True, padding="longest",
                                       max_length=max_length, return_tensors="pt"))
        if images is not None:
            data["pixel_values"] = (torch.stack([self._one(i) for i in images]) if isinstance(images, (list, tuple))
                                    else self._one(images))
        try:
            from transformers import BatchFeature
            return BatchFeature(data=data)
        except Exception:                                   # transformers not importable: same mapping + attribute access
            return _Encoding(**data)


# --------------------------------------------------------------------------------------------------
# modules
# --------------------------------------------------------------------------------------------------
class _EngineModule(nn.Module):
    """nn.Module shell so .eval()/.cuda()/.to() calls of existing callers keep working; holds no
    parameters (weights live repacked inside the engine)."""

    def __init__(self, engine: HipEngine):
        super().__init__()
        object.__setattr__(self, "_engine", engine)


class SiglipProcessor:
    """What `AutoProcessor.from_pretrained("google/siglip-large-patch16-384")` does to images (image_encoder.py:45-48):
    HF SiglipImageProcessor -- convert to RGB (alpha dropped), stretch to size x size with Pillow's BICUBIC resampler,
    rescale by 1/255, normalise with mean = std = 0.5.  Runs on the engine's GPU (`sv_preprocess_image`, recipe 1; bit
    identical to the HF PIL processor); call signature and return shape follow the HF processor."""

    class _Out:
        def __init__(self, pixel_values):
            self.pixel_values = pixel_values

        def __getitem__(self, k):
            return getattr(self, k)

    def __init__(self, size: int = 384, device=None, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)):
        self.size, self.device, self.mean, self.std = size, device, tuple(mean), tuple(std)

    def __call__(self, images=None, return_tensors="pt", **kw):
        import numpy as np
        from .engine import op_preprocess_images
        if images is None:
            raise ValueError("images is required")
        imgs = images if isinstance(images, (list, tuple)) else [images]
        px = []
        for img in imgs:
            if img.mode not in ("RGB", "RGBA"):
                img = img.convert("RGB")
            px.append(torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).to(self.device, non_blocking=True))
        return SiglipProcessor._Out(op_preprocess_images(px, self.size, self.mean, self.std, recipe="siglip"))   # one call


class ImageEncoder(_EngineModule):
    """image_encoder.py:91-94 (clip branch): ln_vision(VisionTransformer(image)) -> [B,257,1024]; siglip branch
    (:108-109): the HF vision tower's last_hidden_state -> [B,576,1024]."""

    def __init__(self, engine: HipEngine, image_size: int = 224, image_encoder_type: str = "clip"):
        super().__init__(engine)
        self.image_encoder_type = image_encoder_type
        # the engine lives on a GPU, so the images are pre-processed there as well (bit-identical to the Pillow recipe)
        dev = torch.device("cuda", engine.device) if engine is not None and hasattr(engine, "device") else None
        if "siglip" in image_encoder_type:
            self.processor = SiglipProcessor(size=image_size, device=dev)
        else:
            self.processor = ImageTrainProcessor(size=image_size, device=dev)

    def forward(self, image: torch.Tensor) -> torch.Tensor:
        return self._engine.encode_image(image)

    def process_images(self, images):                      # image_encoder.py:112-117
        if self.image_encoder_type == "clip":
            if hasattr(self.processor, "batch") and len(images) > 1:        # same list of [1, 3, S, S] tensors, one device call
                return list(self.processor.batch(list(images)).unsqueeze(1).unbind(0))
            return [self.processor(image).unsqueeze(0) for image in images]
        return self.processor(images=images, return_tensors="pt").pixel_values.unsqueeze(0)     # sic: [1, B, 3, S, S]


class Adapter(_EngineModule):
    """adapter.py:33-39: Linear -> Swish -> Linear -> LayerNorm([Q,D]) | BatchNorm1d(Q)."""

    def __init__(self, engine: HipEngine, query_length: int, adapter_norm: str):
        super().__init__(engine)
        self.query_length = query_length
        self.adapter_norm = adapter_norm

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        return self._engine.adapter(hidden_states)


class _TokenEmbedding(_EngineModule):
    def forward(self, input_ids: torch.Tensor) -> torch.Tensor:
        return self._engine.embed_tokens(input_ids)


class _Backbone(_EngineModule):
    """stand-in for GPTBigCodeModel / Starcoder2Model: exposes ``wte`` (starvector_v1.py:16-18) and
    ``embed_tokens`` (starvector_v2.py:45-47)."""

    def __init__(self, engine: HipEngine):
        super().__init__(engine)
        self.wte = _TokenEmbedding(engine)
        self.embed_tokens = self.wte


class HipCausalLM(_EngineModule):
    """The object the reference reaches as ``svg_transformer.transformer``: ``.generate(**kwargs)`` with
    the keyword set built at starvector_base.py:228-241 (+ :289-295), ``.transformer.wte``."""

    def __init__(self, engine: HipEngine, eos_token_id: int, pad_token_id: int):
        super().__init__(engine)
        self.transformer = _Backbone(engine)          # GPTBigCodeForCausalLM.transformer
        self.model = self.transformer                 # Starcoder2ForCausalLM.model
        self.eos_token_id = eos_token_id
        self.pad_token_id = pad_token_id
        self.seed = None          # None: every sampling call draws its seed from torch's generator; an int pins it
        self.batcher = None       # a batching.ContinuousBatcher: concurrent single-sequence calls share one decode loop

    @staticmethod
    def _stop_ids(stopping_criteria) -> Optional[List[int]]:
        if not stopping_criteria:
            return None
        stops: List[List[int]] = []
        for crit in stopping_criteria:
            s = getattr(crit, "stops", None)
            if s is None:
                raise NotImplementedError(f"unsupported stopping criterion {type(crit).__name__}: the engine evaluates "
                                          "the reference's StoppingCriteriaSub (token-sequence stop) on device")
            stops.extend([list(map(int, x)) for x in s])
        if len(stops) > 1:
            raise NotImplementedError("one stop sequence is supported (the reference passes exactly one: '</svg>')")
        return stops[0] if stops else None

    def _generate_padded(self, inputs_embeds, attention_mask, kw):
        mask = attention_mask.to(torch.bool)
        B, S, _ = inputs_embeds.shape
        if not bool(mask[:, -1].all()):
            raise ValueError("the last prompt position of every row must be a real token (generation continues from it)")
        budget = int(kw["max_length"]) - S                         # HF counts the budget from the PADDED prompt length
        if budget <= 0:
            raise ValueError(f"max_length ({kw['max_length']}) must exceed the prompt length ({S})")
        lengths = mask.sum(dim=1).tolist()
        pad = int(self.pad_token_id if kw.get("pad_token_id") is None else kw["pad_token_id"])
        stop = self._stop_ids(kw.get("stopping_criteria"))
        eng = self._engine
        if (int(kw.get("num_beams") or 1) == 1 and hasattr(eng, "cb_admit") and B <= eng.cfg.max_batch
                and (getattr(self, "batcher", None) is None or _in_exclusive_job())):
            return self._generate_padded_slots(inputs_embeds, mask, lengths, budget, pad, stop, kw)
        outs, fired_at = [None] * B, None
        for n in sorted(set(lengths)):
            rows = [b for b in range(B) if lengths[b] == n]
            emb = torch.stack([inputs_embeds[b][mask[b]] for b in rows], 0)
            # HF subtracts the PADDED prompt length from min_length as well: the same number of EOS-free steps for every row
            sub = dict(kw, max_length=n + budget, min_length=n + max(int(kw.get("min_length") or 0) - S, 0))
            if rows[0] != 0:
                sub["stopping_criteria"] = None                    # the reference's stop looks at row 0 of the batch only
            toks = self.generate(inputs_embeds=emb, attention_mask=None, **sub)
            if rows[0] == 0 and stop and toks.shape[1] >= len(stop) and toks[0, -len(stop):].tolist() == list(stop):
                fired_at = toks.shape[1]                           # row 0 ended the whole batch at this step
            for i, b in enumerate(rows):
                outs[b] = toks[i]
        L = fired_at if fired_at is not None else max(t.shape[0] for t in outs)
        res = torch.full((B, L), pad, dtype=torch.long, device=inputs_embeds.device)
        for b, t in enumerate(outs):
            k = min(L, t.shape[0])
            res[b, :k] = t[:k]
        return res

    def _generate_padded_slots(self, inputs_embeds, mask, lengths, budget, pad, stop, kw):
        """Rows of different real length in ONE decode loop: every row is a slot of the engine's continuous batch (own prompt
        length, positions and KV pages), prompt passes run per length group (they are rectangular), and all rows then decode
        together -- instead of one full generate call per length group.  HF semantics of the padded batch are restored on
        the host: the reference's row-0 stop ends every row at row 0's step, finished rows are padded."""
        eng = self._engine
        B, S, _ = inputs_embeds.shape
        eos = int(self.eos_token_id if kw.get("eos_token_id") is None else kw["eos_token_id"])
        min_new = max(int(kw.get("min_length") or 0) - S, 0)
        seed = int(kw.get("seed") or 0)
        base = dict(max_new_tokens=budget, do_sample=bool(kw.get("do_sample")), eos_token_id=eos, pad_token_id=pad,
                    temperature=float(kw.get("temperature") if kw.get("temperature") is not None else 1.0),
                    top_p=float(kw.get("top_p") if kw.get("top_p") is not None else 1.0), top_k=int(kw.get("top_k") or 0),
                    repetition_penalty=float(kw.get("repetition_penalty") or 1.0), min_new_tokens=min_new)
        slot_of = [None] * B
        lock = getattr(eng, "call_lock", None) or contextlib.nullcontext()
        with lock:            # from the first cb_reset to the last: another thread's generate must not reset or admit in between
            outs, fired_len = self._run_slots(eng, inputs_embeds, mask, lengths, base, seed, stop, slot_of)
        L = fired_len if fired_len is not None else max(t.shape[0] for t in outs)
        res = torch.full((B, L), pad, dtype=torch.long, device=inputs_embeds.device)
        for b, t in enumerate(outs):
            k = min(L, t.shape[0])
            res[b, :k] = t[:k].to(res.device)
        return res

    @staticmethod
    def _run_slots(eng, inputs_embeds, mask, lengths, base, seed, stop, slot_of):
        B = inputs_embeds.shape[0]
        eng.cb_reset()
        try:
            for n in sorted(set(lengths)):
                rows = [b for b in range(B) if lengths[b] == n]
                emb = torch.stack([inputs_embeds[b][mask[b]] for b in rows], 0).to(torch.bfloat16).contiguous()
                reqs = [dict(base, seed=(seed + 0x9E3779B97F4A7C15 * b) & (2 ** 63 - 1), stop_ids=stop if b == 0 else None)
                        for b in rows]
                for b, s_ in zip(rows, eng.cb_admit(emb, reqs)):
                    slot_of[b] = s_
            fired_len = None
            while True:
                live = eng.cb_step(16)
                lv, st = eng.cb_poll()
                if stop and not lv[slot_of[0]]:
                    r0 = eng.cb_read(slot_of[0], 0, st[slot_of[0]]).tolist()
                    if len(r0) >= len(stop) and r0[-len(stop):] == list(stop):
                        fired_len = len(r0)            # row 0 ended the whole batch at this step (starvector_base.py:9-20)
                        break
                if live == 0:
                    break
            lv, st = eng.cb_poll()
            outs = [eng.cb_read(slot_of[b], 0, st[slot_of[b]]) for b in range(B)]
        finally:
            eng.cb_reset()
        return outs, fired_len

    @torch.no_grad()
    def generate(self, inputs_embeds: torch.Tensor = None, attention_mask: Optional[torch.Tensor] = None,
                 do_sample: bool = False, top_p: Optional[float] = 1.0, temperature: Optional[float] = 1.0,
                 num_beams: int = 1, max_length: int = 30, min_length: int = 0, repetition_penalty: float = 1.0,
                 length_penalty: float = 1.0, use_cache: bool = True, stopping_criteria=None,
                 early_stopping: bool = False, pad_token_id: Optional[int] = None, eos_token_id: Optional[int] = None,
                 num_return_sequences: int = 1, top_k: Optional[int] = 50, streamer=None, seed: Optional[int] = None,
                 **unused) -> torch.Tensor:
        # top_k: the reference never passes it; its pinned transformers==4.49.0 (pyproject.toml:18) defaults
        # GenerationConfig.top_k to 50, so every do_sample call there is top-k 50 followed by top-p.  Same default here.
        if inputs_embeds is None:
            raise ValueError("inputs_embeds is required (the reference always generates from embeddings)")
        # Random stream: HF draws from torch's global generator, so repeated calls differ and torch.manual_seed controls them.
        # The device sampler is a pure function of (seed, step, row): the per-call seed is therefore DRAWN from torch's
        # generator (same reproducibility contract), unless the caller pins it with `seed=` (tests, data-parallel ranks).
        if seed is None:
            seed = self.seed
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, ()).item()) if do_sample else 0
        seed = int(seed) & (2 ** 63 - 1)
        num_beams = int(num_beams)
        if num_beams < 1:
            raise ValueError("`num_beams` has to be an integer strictly greater than 0")     # HF's own check
        if num_beams > 8:
            raise NotImplementedError("num_beams > 8 is not built")
        num_return_sequences = int(num_return_sequences or 1)
        if num_return_sequences < 1:
            raise ValueError("`num_return_sequences` has to be at least 1")
        if num_return_sequences > 1:
            # HF expands the inputs (repeat_interleave) and samples every copy independently; the reference forces
            # num_beams = 1 whenever it asks for several sequences (starvector_base.py:273-276)
            if num_beams > 1:
                raise NotImplementedError("num_return_sequences > 1 with beam search is not built "
                                          "(the reference sets num_beams=1 on this path)")
            inputs_embeds = inputs_embeds.repeat_interleave(num_return_sequences, dim=0)
            attention_mask = None if attention_mask is None else attention_mask.repeat_interleave(num_return_sequences, dim=0)
        if repetition_penalty is not None and not repetition_penalty > 0:
            raise ValueError("`repetition_penalty` has to be a strictly positive float")   # HF's own check
        if attention_mask is not None and not bool((attention_mask == 1).all()):
            # Padded prompts (text2svg with captions of different lengths; the v2 tokenizer pads on the left, llm/starcoder2.py:53).
            # HF masks the padded keys and numbers positions by cumsum(mask), so a padded row behaves exactly like the same row
            # with its padding removed (checked against HF for left and right padding, GPTBigCode and StarCoder2).  The engine
            # takes rectangular all-ones prompts, so rows are grouped by their real length and generated group by group.
            return self._generate_padded(inputs_embeds, attention_mask, dict(
                do_sample=do_sample, top_p=top_p, temperature=temperature, num_beams=num_beams, max_length=max_length,
                min_length=min_length, repetition_penalty=repetition_penalty, length_penalty=length_penalty,
                use_cache=use_cache, stopping_criteria=stopping_criteria, early_stopping=early_stopping,
                pad_token_id=pad_token_id, eos_token_id=eos_token_id, top_k=top_k, seed=seed))
        S0 = inputs_embeds.shape[1]
        # HF (_prepare_generated_length): with inputs_embeds min_length is reduced by the prompt length -- 0 on the im2svg path
        # (SURVEY.md 8a-a11), positive only for a text2svg caption shorter than min_length; MinLengthLogitsProcessor then
        # keeps EOS at -inf for that many new tokens (in beam search HF applies it to the log-probabilities: so does the scorer).
        min_new = max(int(min_length or 0) - S0, 0)
        if not use_cache:
            pass        # the engine always uses its paged KV cache; results are identical
        on_tokens = None
        if streamer is not None:
            # HF streamer protocol (generation/streamers.py): put(prompt ids) once, put(next_tokens [B]) per step, end().
            # Tokens arrive in bursts of `sync_every` steps: the decode loop does not return to the host every token.
            if num_beams > 1:
                raise ValueError("`streamer` cannot be used with beam search")          # HF's own check
            streamer.put(torch.empty(inputs_embeds.shape[0], 0, dtype=torch.long))     # no prompt ids with inputs_embeds

            def on_tokens(tokens, first_col):
                for c in range(tokens.shape[1]):
                    streamer.put(tokens[:, c])
        batcher = getattr(self, "batcher", None)
        if batcher is not None and not (num_beams == 1 and inputs_embeds.shape[0] == 1) and not _in_exclusive_job():
            # beam search / a multi-row HF batch while requests share the engine: run it with the engine to itself, in turn
            def call():
                _EXCLUSIVE.active = True          # thread-local: only the scheduler thread running this job sees it
                try:
                    return self.generate(inputs_embeds=inputs_embeds, attention_mask=attention_mask, do_sample=do_sample,
                                         top_p=top_p, temperature=temperature, num_beams=num_beams, max_length=max_length,
                                         min_length=min_length, repetition_penalty=repetition_penalty,
                                         length_penalty=length_penalty, use_cache=use_cache, stopping_criteria=stopping_criteria,
                                         early_stopping=early_stopping, pad_token_id=pad_token_id, eos_token_id=eos_token_id,
                                         top_k=top_k, streamer=streamer, seed=seed)
                finally:
                    _EXCLUSIVE.active = False
            return batcher.run_exclusive(call)
        if batcher is not None and num_beams == 1 and inputs_embeds.shape[0] == 1 and not _in_exclusive_job():
            # serving: one request per call (serve/model_worker.py:120-181), many calls in flight -> they share the engine's
            # decode loop instead of taking turns; the tokens are those of the solo call below
            def on_chunk(toks, first):
                if on_tokens is not None:
                    on_tokens(toks.view(1, -1), first)
            out = self.batcher.generate(inputs_embeds.to(torch.bfloat16), dict(
                max_new_tokens=int(max_length) - S0, do_sample=bool(do_sample),
                temperature=float(temperature if temperature is not None else 1.0),
                top_p=float(top_p if top_p is not None else 1.0), top_k=int(top_k or 0), seed=seed,
                eos_token_id=int(self.eos_token_id if eos_token_id is None else eos_token_id),
                pad_token_id=int(self.pad_token_id if pad_token_id is None else pad_token_id),
                stop_ids=self._stop_ids(stopping_criteria),
                repetition_penalty=float(repetition_penalty if repetition_penalty is not None else 1.0),
                min_new_tokens=min_new), on_chunk if on_tokens is not None else None).to(inputs_embeds.device)
            if streamer is not None:
                streamer.end()
            return out
        lock = getattr(self._engine, "call_lock", None) or contextlib.nullcontext()
        with lock:          # the same lock the slot path holds: a classic generate never lands between its cb_reset / cb_admit
            out = self._classic_generate(inputs_embeds, max_length, do_sample, temperature, top_p, eos_token_id, pad_token_id,
                                         stopping_criteria, seed, repetition_penalty, num_beams, length_penalty, early_stopping,
                                         top_k, on_tokens, streamer, min_new)
        if streamer is not None:
            streamer.end()
        return out

    def _classic_generate(self, inputs_embeds, max_length, do_sample, temperature, top_p, eos_token_id, pad_token_id,
                          stopping_criteria, seed, repetition_penalty, num_beams, length_penalty, early_stopping, top_k,
                          on_tokens, streamer, min_new):
        return self._engine.generate(
            inputs_embeds.to(torch.bfloat16), max_length=int(max_length), do_sample=bool(do_sample),
            temperature=float(temperature if temperature is not None else 1.0),
            top_p=float(top_p if top_p is not None else 1.0),
            eos_token_id=int(self.eos_token_id if eos_token_id is None else eos_token_id),
            pad_token_id=int(self.pad_token_id if pad_token_id is None else pad_token_id),
            stop_ids=self._stop_ids(stopping_criteria), seed=seed,
            repetition_penalty=float(repetition_penalty if repetition_penalty is not None else 1.0),
            num_beams=num_beams, length_penalty=float(length_penalty if length_penalty is not None else 1.0),
            early_stopping=early_stopping, top_k=int(top_k or 0), on_tokens=on_tokens,
            sync_every=8 if streamer is not None else 32, **({"min_new_tokens": min_new} if min_new else {}))


class StoppingCriteriaSub:
    """starvector_base.py:9-20: stop the whole batch when ROW 0 ends with one of ``stops``.
    Carried to the device by HipCausalLM.generate (no per-step host sync)."""

    def __init__(self, stops=()):
        self.stops = [list(s) for s in stops]

    def __call__(self, input_ids, scores=None, **kw):
        return any(input_ids[0][-len(s):].tolist() == s for s in self.stops)


class StarCoderModel(nn.Module):
    """llm/starcoder.py:9-53: tokenizer + causal LM + the '<svg' prompt."""

    def __init__(self, engine: HipEngine, tokenizer, max_length: int, v2: bool = False):
        super().__init__()
        self.tokenizer = tokenizer
        self.max_length = max_length
        # v1 passes pad_token_id=[PAD] explicitly (starvector_base.py:289-295); v2 passes nothing
        # (starvector_v2.py:53-57), so HF falls back to pad_token_id = eos_token_id
        self.transformer = HipCausalLM(engine, tokenizer.eos_token_id,
                                       tokenizer.eos_token_id if v2 else tokenizer.pad_token_id)
        self.prompt = "<svg"
        self.svg_start_token = "<svg-start>"
        self.image_start_token = "<image-start>"
        self.text_start_token = "<caption-start>"
        self.svg_start_token_id = tokenizer.encode(self.svg_start_token)[0]
        if v2:
            self.svg_end_token = "<svg-end>"
            self.svg_end_token_id = tokenizer.encode(self.svg_end_token)[0]


class StarVectorStarCoder(nn.Module):
    """StarVectorBase + v1 binding (starvector_base.py:22-48,203-295; starvector_v1.py)."""

    def __init__(self, config: StarVectorConfig, engine: HipEngine, tokenizer, v2: bool = False):
        super().__init__()
        self.task = "im2svg"
        self.model_precision = torch.bfloat16
        ec = engine.cfg
        self.svg_transformer = StarCoderModel(engine, tokenizer, config.max_length, v2=v2)
        self.image_encoder = ImageEncoder(engine, ec.image_size, config.image_encoder_type)
        self.query_length = ec.query_length                          # starvector_base.py:85-106
        self.image_projection = Adapter(engine, self.query_length, ec.adapter_norm)
        self.max_length = config.max_length_train - self.query_length - 4
        # starvector_v1.py:10 -> AutoProcessor = SimpleStarVectorProcessor (HF-style call); starvector_v2.py:12 -> the tower's
        # image processor.  `image_encoder.processor` stays the plain callable `process_images` uses (image_encoder.py:112-117)
        self.processor = self.image_encoder.processor if v2 else SimpleStarVectorProcessor(
            tokenizer, size=ec.image_size, device=getattr(self.image_encoder.processor, "device", None))

    def use_image_encoder(self):
        return True

    def _get_embeddings(self, input_ids):                             # starvector_v1.py:16-18
        return self.svg_transformer.transformer.transformer.wte(input_ids)

    def _tokenize(self, text, max_length, device, add_special_tokens=True):
        tokens = self.svg_transformer.tokenizer(text, add_special_tokens=add_special_tokens, padding="longest",
                                                return_tensors="pt")
        return tokens.to(device)

    def _prepare_generation_inputs(self, batch, prompt, device):      # starvector_base.py:203-221
        image = batch["image"].to(device).to(self.model_precision)
        if image.dim() == 3:                  # one un-batched image, what `processor(pil)["pixel_values"]` returns for v1
            image = image.unsqueeze(0)
        if prompt is None:
            prompt = self.svg_transformer.prompt
        prompt_tokens = self._tokenize([prompt] * image.size(0), None, device, add_special_tokens=False)
        # image_projection(image_encoder(image)) and _get_embeddings(prompt ids), written straight into ONE inputs_embeds buffer instead of
        # two tensors + torch.cat (same values: tests/test_gpu_e2e.py; the separate module calls stay available and equal)
        enc = self.image_encoder(image)
        ids = prompt_tokens.input_ids.to(device)
        prep = getattr(getattr(self.image_projection, "_engine", None), "prepare_inputs", None)
        if prep is not None and not bool((prompt_tokens.attention_mask == 0).any()):
            inputs_embeds = prep(enc, ids)
        else:
            inputs_embeds = torch.cat([self.image_projection(enc), self._get_embeddings(ids)], dim=1)
        embedded_att = torch.ones(inputs_embeds.shape[0], inputs_embeds.shape[1] - ids.shape[1], dtype=torch.long, device=device)
        attention_mask = torch.cat([embedded_att, prompt_tokens.attention_mask], dim=1)
        return inputs_embeds, attention_mask, prompt_tokens

    def _get_generation_kwargs(self, base_kwargs):                    # starvector_base.py:223-241
        end_sequence = self.svg_transformer.tokenizer("</svg>", add_special_tokens=False)["input_ids"]
        return {
            "inputs_embeds": base_kwargs["inputs_embeds"],
            "attention_mask": base_kwargs["attention_mask"],
            "do_sample": base_kwargs.get("use_nucleus_sampling", True),
            "top_p": base_kwargs.get("top_p", 0.9),
            "temperature": base_kwargs.get("temperature", 1),
            "num_beams": base_kwargs.get("num_beams", 2),
            "max_length": base_kwargs.get("max_length", 30),
            "min_length": base_kwargs.get("min_length", 1),
            "repetition_penalty": base_kwargs.get("repetition_penalty", 1.0),
            "length_penalty": base_kwargs.get("length_penalty", 1.0),
            "use_cache": base_kwargs.get("use_cache", True),
            "stopping_criteria": [StoppingCriteriaSub(stops=[end_sequence])],
            # not in the reference's whitelist (so its serve worker's streamer never streams, serve/model_worker.py:129-175);
            # kept here so that the same call does stream
            "streamer": base_kwargs.get("streamer"),
            # extension: pins the device sampler's random stream (default: drawn per call from torch's generator, so that
            # `torch.manual_seed` reproduces a run and repeated calls differ, as with HF's torch.multinomial)
            "seed": base_kwargs.get("seed"),
        }

    def _get_im2svg_specific_kwargs(self, kwargs):                    # starvector_base.py:289-295
        return {"early_stopping": True, "pad_token_id": self.svg_transformer.tokenizer.pad_token_id}

    def _get_text2svg_specific_kwargs(self, kwargs):                  # starvector_base.py:332-339
        return {"eos_token_id": self.svg_transformer.tokenizer.eos_token_id, "early_stopping": True,
                "length_penalty": kwargs.get("length_penalty", 1.0)}

    def generate_text2svg(self, batch, **kwargs):
        """starvector_base.py:297-330, restated by intent: embed(caption ids + <svg-start>) -> generate -> new token ids.
        The snapshot's version cannot run (it passes two positional arguments to _get_generation_kwargs, :321-324, and
        subtracts the prompt length from max_length twice); here `max_length` counts the prompt once, as in im2svg.
        Captions of different lengths are padded by the tokenizer; the padded rows are generated group by group
        (HipCausalLM._generate_padded), which reproduces HF's masked generation exactly."""
        device = batch["image"].device if "image" in batch else torch.device("cuda", self._engine_device())
        prompt_tokens = self._tokenize(list(batch["caption"]), kwargs.get("max_length", 30), device, add_special_tokens=False)
        trigger = self._tokenize([self.svg_transformer.svg_start_token] * len(batch["caption"]), None, device,
                                 add_special_tokens=False)
        input_tokens = torch.cat([prompt_tokens.input_ids, trigger.input_ids], dim=1)
        attention_mask = torch.cat([prompt_tokens.attention_mask, trigger.attention_mask], dim=1)
        inputs_embeds = self._get_embeddings(input_tokens)
        generation_kwargs = self._get_generation_kwargs({**kwargs, "inputs_embeds": inputs_embeds,
                                                         "attention_mask": attention_mask})
        generation_kwargs.update(self._get_text2svg_specific_kwargs(kwargs))
        return self.svg_transformer.transformer.generate(**generation_kwargs)

    def _engine_device(self):
        return self.svg_transformer.transformer._engine.device

    def generate_im2svg(self, batch, **kwargs):                       # starvector_base.py:243-259
        return self.generate_im2svg_grpo(batch, **kwargs)["raw_svg"]

    def generate_im2svg_grpo(self, batch, **kwargs):                  # starvector_base.py:261-286
        device = batch["image"].device
        inputs_embeds, attention_mask, prompt_tokens = self._prepare_generation_inputs(batch, kwargs.get("prompt"), device)
        generation_kwargs = self._get_generation_kwargs({**kwargs, "inputs_embeds": inputs_embeds,
                                                         "attention_mask": attention_mask})
        generation_kwargs.update(self._get_im2svg_specific_kwargs(kwargs))
        num_return_sequences = kwargs.get("num_return_sequences", 1)
        if num_return_sequences > 1:                                   # :273-276
            generation_kwargs["num_return_sequences"] = num_return_sequences
            generation_kwargs["num_beams"] = 1
        outputs = self.svg_transformer.transformer.generate(**generation_kwargs)
        outputs = torch.cat([prompt_tokens.input_ids.repeat(num_return_sequences, 1), outputs], dim=1)    # :279
        raw_svg = self.svg_transformer.tokenizer.batch_decode(outputs, skip_special_tokens=True)
        return {"raw_svg": raw_svg, "outputs": outputs, "inputs_embeds": inputs_embeds}


class StarVectorStarCoder2(StarVectorStarCoder):
    """v2 binding (starvector_v2.py:8-63): StarCoder2 decoder, SigLIP tower, `embed_tokens`, no im2svg kwargs."""

    def __init__(self, config: StarVectorConfig, engine: HipEngine, tokenizer):
        super().__init__(config, engine, tokenizer, v2=True)

    def _get_embeddings(self, input_ids):                             # starvector_v2.py:45-47
        return self.svg_transformer.transformer.model.embed_tokens(input_ids)

    def _get_im2svg_specific_kwargs(self, kwargs):                    # starvector_v2.py:53-57
        return {}

    def _get_text2svg_specific_kwargs(self, kwargs):                  # starvector_v2.py:59-63
        return {"eos_token_id": self.svg_transformer.tokenizer.eos_token_id}


class StarVectorForCausalLM(nn.Module):
    """starvector_arch.py:133-193 facade."""
    config_class = StarVectorConfig

    def __init__(self, config: StarVectorConfig, state_dict: Optional[Dict[str, torch.Tensor]] = None, tokenizer=None,
                 device: Optional[int] = None):
        super().__init__()
        self.config = config
        self.engine = HipEngine(config.engine_config(), device=device)
        if state_dict is not None:
            self.engine.load_state_dict(state_dict)
        if tokenizer is None:
            tokenizer = ByteTokenizer(config.vocab_size, v2=config.is_v2)
        cls = StarVectorStarCoder2 if config.is_v2 else StarVectorStarCoder       # starvector_arch.py:139-144
        self.model = cls(config, self.engine, tokenizer)

    @classmethod
    def from_pretrained(cls, path: str, torch_dtype="bfloat16", tokenizer=None, **kwargs):
        """Load a LOCAL checkpoint directory in the reference's format (config.json + *.safetensors).
        There is no network here: hub names are rejected with a clear error."""
        if not os.path.isdir(path):
            raise FileNotFoundError(f"{path!r} is not a local directory (hub download is unavailable offline)")
        with open(os.path.join(path, "config.json")) as f:
            cfg_json = json.load(f)
        from safetensors.torch import load_file
        sd: Dict[str, torch.Tensor] = {}
        for fn in sorted(os.listdir(path)):
            if fn.endswith(".safetensors"):
                sd.update(load_file(os.path.join(path, fn)))
        cfg = config_from_checkpoint({**cfg_json, **{k: v for k, v in kwargs.items() if k != "byte_tokenizer_fallback"}},
                                     {k: tuple(v.shape) for k, v in sd.items()})
        if tokenizer is None:
            tok_files = ("tokenizer.json", "tokenizer_config.json", "vocab.json", "tokenizer.model", "merges.txt")
            if any(os.path.exists(os.path.join(path, f)) for f in tok_files):
                # a checkpoint that ships a tokenizer must load it: a silent byte-level stand-in would turn '<svg' / '</svg>'
                # into the wrong ids (the stop would never fire, the output would decode to garbage) -- errors propagate
                from transformers import AutoTokenizer
                tokenizer = AutoTokenizer.from_pretrained(path, use_fast=False)
            elif kwargs.get("byte_tokenizer_fallback", False):
                import warnings
                warnings.warn(f"{path!r} has no tokenizer files: using the byte-level stand-in tokenizer (ids are bytes + 1; "
                              "NOT the StarCoder vocabulary -- only meaningful for synthetic weights)", RuntimeWarning)
                tokenizer = None
            else:
                raise FileNotFoundError(
                    f"{path!r} has no tokenizer files (tokenizer.json / vocab.json / ...): pass tokenizer=..., or "
                    "byte_tokenizer_fallback=True to use the byte-level stand-in (synthetic weights only)")
        return cls(cfg, state_dict=sd, tokenizer=tokenizer)

    @torch.no_grad()
    def forward(self, vision_embeds, input_ids, num_generations, attention_mask, num_logits_to_keep):
        """starvector_arch.py:161-184, inference mode (no autograd graph: the engine holds no torch parameters): logits of
        the completions given the visual prefix, as GRPO's log-prob pass asks for them."""
        completion_embeds = self.model._get_embeddings(input_ids)
        inputs_embeds = torch.cat([vision_embeds.repeat(num_generations, 1, 1).to(completion_embeds.dtype),
                                   completion_embeds], dim=1)
        lead = None
        if attention_mask is not None and not bool((attention_mask == 1).all()):
            # Right padding (completions padded after their EOS, what a GRPO trainer passes): under the causal mask a real
            # position never sees a later key and HF's positions (cumsum(mask) - 1) equal the plain index for it, so the
            # logits of every real position are those of the unmasked run.  Rows at padded positions are unspecified (HF's
            # differ there too from any unpadded run; the caller multiplies them away with the same mask).
            # Left padding: the pinned transformers (4.49; gpt_bigcode/modeling_gpt_bigcode.py:980-983) masks the padded keys and
            # numbers positions by cumsum(mask) - 1 inside the model's forward, i.e. a left-padded row IS the same row with its
            # padding removed (pinned against HF: oracle/make_golden.py::run_forward_case, tests/golden/tiny_forward) -> rows are
            # grouped by their number of leading pads, scored without them, and their logits put back at the real positions.
            m = attention_mask.to(torch.bool)
            if m.shape != inputs_embeds.shape[:2]:
                raise ValueError(f"attention_mask {tuple(m.shape)} does not cover inputs_embeds {tuple(inputs_embeds.shape[:2])}")
            lead = (m.cumsum(1) == 0).sum(1)                                  # leading pads per row
            if bool((lead >= m.shape[1]).any()):
                empty = (lead >= m.shape[1]).nonzero().flatten().tolist()
                raise ValueError(f"attention_mask rows {empty} are all zeros: a row needs at least one real position to be scored")
            idx = torch.arange(m.shape[1], device=m.device).unsqueeze(0)
            real = idx >= lead.unsqueeze(1)
            if bool((m[:, 1:] & ~m[:, :-1] & real[:, :-1]).any()):            # a 1 after a 0 behind the leading pads
                raise NotImplementedError("attention masks with holes are not built for the scoring forward (left / right padding only)")
            if not bool(lead.any()):
                lead = None
        emb16, keep = inputs_embeds.to(torch.bfloat16), int(num_logits_to_keep or 0)
        lm = getattr(getattr(self.model, "svg_transformer", None), "transformer", None)
        batcher = getattr(lm, "batcher", None)

        def score(e16):
            if batcher is not None and not _in_exclusive_job():
                # requests share the engine's decode loop: the scoring pass wants the engine to itself (it would fail with SV_ESTATE
                # while slots are live) -> queue it as an exclusive job, FIFO with the generation requests
                def call():
                    _EXCLUSIVE.active = True
                    try:
                        return self.engine.forward_logits(e16, keep)
                    finally:
                        _EXCLUSIVE.active = False
                return batcher.run_exclusive(call)
            with (getattr(self.engine, "call_lock", None) or contextlib.nullcontext()):
                return self.engine.forward_logits(e16, keep)

        if lead is None:
            logits = score(emb16)
        else:
            S = emb16.shape[1]
            if keep and int(lead.max()) + keep > S:
                raise ValueError(f"num_logits_to_keep ({keep}) reaches into the left padding of a row ({int(lead.max())} pads of {S})")
            logits = None
            for pads in sorted(set(lead.tolist())):
                rows = (lead == pads).nonzero().flatten()
                lg = score(emb16[rows, pads:].contiguous())                   # [rows, keep or S - pads, V]
                if logits is None:
                    logits = torch.zeros(emb16.shape[0], keep or S, lg.shape[-1], dtype=lg.dtype, device=lg.device)
                logits[rows.to(lg.device), (0 if keep else pads):] = lg
        try:
            from transformers.modeling_outputs import CausalLMOutputWithCrossAttentions
            return CausalLMOutputWithCrossAttentions(loss=None, logits=logits, past_key_values=None, hidden_states=None,
                                                     attentions=None, cross_attentions=None)
        except Exception:                                   # transformers not importable: same field names
            import types
            return types.SimpleNamespace(loss=None, logits=logits, past_key_values=None, hidden_states=None,
                                         attentions=None, cross_attentions=None)

    def generate_im2svg(self, batch, **kwargs):           # starvector_arch.py:186-187
        return self.model.generate_im2svg(batch, **kwargs)

    def generate_im2text(self, batch, **kwargs):          # starvector_arch.py:189-190 forwards to a method the
        raise NotImplementedError("generate_im2text does not exist in the reference either (starvector_arch.py:189)")

    def process_images(self, images):                     # starvector_arch.py:192-193
        return self.model.image_encoder.process_images(images)
