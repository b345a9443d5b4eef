"""CPU: host-side mirror of the reference interface (no GPU compute involved)."""
import os

import pytest
import torch

import starvector_amd as sva
from starvector_amd.model import ByteTokenizer, ImageTrainProcessor, StarVectorConfig, StoppingCriteriaSub, HipCausalLM
from starvector_amd.parallel import shard_bounds, shard_batch


def test_config_maps_to_engine_shapes():
    ec = StarVectorConfig().engine_config()
    assert (ec.vit_width, ec.vit_layers, ec.vit_heads, ec.hidden, ec.n_layer, ec.n_head, ec.n_inner) == \
        (1024, 23, 16, 2048, 24, 16, 8192)
    assert ec.vocab == 49152 + 4 and ec.query_length == 257          # SURVEY.md section 8 header
    v2 = StarVectorConfig(starcoder_model_name="bigcode/starcoder2-7b", image_encoder_type="siglip_384", hidden_size=4608,
                          num_hidden_layers=32, num_attention_heads=36, num_kv_heads=4, n_inner=18432, added_tokens=5,
                          n_positions=16384, max_length=16000, max_batch=16).engine_config()
    assert (v2.arch, v2.n_kv_head, v2.hidden, v2.n_head, v2.n_inner, v2.vocab) == ("v2", 4, 4608, 36, 18432, 49157)
    assert v2.query_length == 576 and v2.max_seq_len == 16000           # siglip_384: 24x24 patches, no class token
    assert v2.sliding_window == 4097        # visible keys: the reference loads StarCoder2 with FA2, 4.49 passes window_size=(W, W)
    sd = StarVectorConfig(starcoder_model_name="bigcode/starcoder2-7b", image_encoder_type="siglip_384", hidden_size=4608,
                          num_hidden_layers=32, num_attention_heads=36, num_kv_heads=4, n_inner=18432, added_tokens=5,
                          n_positions=16384, max_length=16000, max_batch=16, window_semantics="sdpa").engine_config()
    assert sd.sliding_window == 4096                                    # the eager / sdpa mask shows W keys
    with pytest.raises(NotImplementedError):
        StarVectorConfig(starcoder_model_name="bigcode/starcoder2-7b", image_encoder_type="clip").engine_config()
    with pytest.warns(UserWarning, match="computes in bfloat16"):        # the reference's fp16 configs keep working, converted
        assert StarVectorConfig(torch_dtype="float16").engine_config().hidden == 2048
    with pytest.raises(ValueError):
        StarVectorConfig(torch_dtype="int8").engine_config()


def test_exclusive_device_reaches_the_engine_config(monkeypatch):
    """The deployment knob: "auto" by default since round 6 (sv_config.exclusive_device = 2: the fused decode launches on until one of
    them finds the GPU shared, then off for good with the failed call re-run), on / off by the config field or -- for callers that only change their
    import line -- by SV_EXCLUSIVE_DEVICE=1 / 0 in the environment; an explicit field wins over the environment; both model families carry it."""
    from starvector_amd.engine import _exclusive_code
    monkeypatch.delenv("SV_EXCLUSIVE_DEVICE", raising=False)
    assert StarVectorConfig().engine_config().exclusive_device == "auto"
    assert [_exclusive_code(v) for v in (False, True, "auto", 2, 0, 1, "0", "1", None)] == [0, 1, 2, 2, 0, 1, 0, 1, 0]
    assert StarVectorConfig(exclusive_device=True).engine_config().exclusive_device is True
    monkeypatch.setenv("SV_EXCLUSIVE_DEVICE", "0")
    assert StarVectorConfig().engine_config().exclusive_device is False
    monkeypatch.setenv("SV_EXCLUSIVE_DEVICE", "1")
    assert StarVectorConfig().engine_config().exclusive_device is True
    assert StarVectorConfig(exclusive_device=False).engine_config().exclusive_device is False
    v2 = StarVectorConfig(starcoder_model_name="bigcode/starcoder2-7b", image_encoder_type="siglip_384", hidden_size=4608,
                          num_hidden_layers=32, num_attention_heads=36, num_kv_heads=4, n_inner=18432, added_tokens=5)
    assert v2.engine_config().exclusive_device is True


def test_byte_tokenizer_surface():
    tok = ByteTokenizer(49152)
    assert len(tok) == 49156 and tok.pad_token_id == 49152 and tok.eos_token_id == 0
    enc = tok(["<svg", "<svg"], add_special_tokens=False, return_tensors="pt")
    assert enc.input_ids.shape == (2, 4) and bool((enc.attention_mask == 1).all())
    ids = tok("</svg>", add_special_tokens=False)["input_ids"]
    assert isinstance(ids, list) and len(ids) == 6
    assert tok.batch_decode(torch.tensor([tok.encode("<svg width") + [tok.pad_token_id, 0]])) == ["<svg width"]
    assert tok.encode("<svg-start>") == [49153]
    tok2 = ByteTokenizer(49152, v2=True)                     # llm/starcoder2.py:53: pads on the left
    e2 = tok2(["ab", "abcd"], return_tensors="pt")
    assert e2.attention_mask.tolist() == [[0, 0, 1, 1], [1, 1, 1, 1]] and e2.input_ids[0, :2].tolist() == [tok2.pad_token_id] * 2
    e1 = tok(["ab", "abcd"], return_tensors="pt")
    assert e1.attention_mask.tolist() == [[1, 1, 0, 0], [1, 1, 1, 1]]


def test_image_processor_matches_reference_recipe():
    """data/util.py:40-68: RGBA->white composite, white pad to square, bicubic resize, ToTensor, CLIP normalise."""
    from PIL import Image
    proc = ImageTrainProcessor(size=224)
    img = Image.new("RGBA", (100, 60), (255, 0, 0, 0))               # fully transparent -> white
    x = proc(img)
    assert x.shape == (3, 224, 224)
    white = (1.0 - torch.tensor(sva.model.CLIP_MEAN)) / torch.tensor(sva.model.CLIP_STD)
    torch.testing.assert_close(x[:, 0, 0], white, rtol=0, atol=1e-5)
    torch.testing.assert_close(x[:, 112, 112], white, rtol=0, atol=1e-5)
    # a 224x224 RGB image goes through untouched apart from normalisation
    g = torch.Generator().manual_seed(0)
    arr = (torch.rand(224, 224, 3, generator=g) * 255).to(torch.uint8)
    y = proc(Image.fromarray(arr.numpy(), "RGB"))
    ref = (arr.permute(2, 0, 1).float() / 255 - torch.tensor(sva.model.CLIP_MEAN).view(3, 1, 1)) / \
        torch.tensor(sva.model.CLIP_STD).view(3, 1, 1)
    torch.testing.assert_close(y, ref, rtol=0, atol=1e-6)
    # non-square: padded with white on the short side
    z = proc(Image.new("RGB", (224, 112), (0, 0, 0)))
    torch.testing.assert_close(z[:, 0, 112], white, rtol=0, atol=1e-5)
    assert float(z[0, 112, 112]) < 0


def test_stopping_criteria_sub_is_row0_only():
    crit = StoppingCriteriaSub(stops=[[5, 6]])
    assert crit(torch.tensor([[1, 5, 6], [0, 0, 0]]))
    assert not crit(torch.tensor([[1, 2, 3], [4, 5, 6]]))             # row 1 is never inspected
    assert not crit(torch.tensor([[6]]))                               # shorter than the stop sequence
    assert HipCausalLM._stop_ids([crit]) == [5, 6]
    assert HipCausalLM._stop_ids(None) is None


def test_generate_kwargs_validation_without_gpu():
    lm = HipCausalLM.__new__(HipCausalLM)
    torch.nn.Module.__init__(lm)
    object.__setattr__(lm, "_engine", None)
    lm.eos_token_id, lm.pad_token_id, lm.seed = 0, 1, 0
    emb = torch.zeros(1, 4, 8)
    with pytest.raises(NotImplementedError):
        lm.generate(inputs_embeds=emb, num_beams=9, max_length=8)
    with pytest.raises(ValueError):
        lm.generate(inputs_embeds=emb, num_beams=0, max_length=8)
    with pytest.raises(ValueError):
        lm.generate(inputs_embeds=emb, repetition_penalty=0.0, max_length=8)
    with pytest.raises(ValueError):
        lm.generate(max_length=8)


def test_shard_bounds_cover_batch_exactly():
    for n in (0, 1, 7, 32, 256):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    b = {"image": torch.arange(10).view(10, 1), "note": "x"}
    s = shard_batch(b, 1, 4)
    assert s["image"].flatten().tolist() == [3, 4, 5] and s["note"] == "x"


def test_package_never_imports_the_oracle():
    src_dir = os.path.dirname(sva.__file__)
    for fn in os.listdir(src_dir):
        if fn.endswith(".py"):
            text = open(os.path.join(src_dir, fn)).read()
            assert "import oracle" not in text and "from oracle" not in text, fn


class _FakeEngine:
    """Stands in for HipEngine on CPU: the 'model' emits, for every row, tokens derived from the row's real prompt only
    (sum of its embedding entries), so host-side batching logic can be checked without a GPU."""
    device = 0

    def __init__(self):
        self.calls = []

    def generate(self, inputs_embeds, max_length, stop_ids=None, eos_token_id=0, pad_token_id=0, on_tokens=None, **kw):
        B, S, _ = inputs_embeds.shape
        budget = max_length - S
        self.calls.append((B, S, budget, stop_ids))
        self.last_kw = kw
        base = inputs_embeds.float().sum(dim=(1, 2)).round().long()
        toks = torch.stack([(base + 3 * t) % 97 + 1 for t in range(budget)], 1)           # [B, budget], never 0
        n = budget
        if stop_ids:                                                                       # row-0 stop, like the engine
            r0 = toks[0].tolist()
            for t in range(len(stop_ids) - 1, budget):
                if r0[t + 1 - len(stop_ids):t + 1] == list(stop_ids):
                    n = t + 1
                    break
        if on_tokens is not None:
            on_tokens(toks[:, :n], 0)
        return toks[:, :n]


def _fake_lm():
    lm = HipCausalLM.__new__(HipCausalLM)
    torch.nn.Module.__init__(lm)
    object.__setattr__(lm, "_engine", _FakeEngine())
    lm.eos_token_id, lm.pad_token_id, lm.seed = 0, 99, 0
    return lm


def test_padded_prompts_are_generated_by_length_groups():
    lm = _fake_lm()
    torch.manual_seed(0)
    emb = torch.randint(0, 5, (4, 6, 3)).float()
    mask = torch.tensor([[0, 0, 1, 1, 1, 1], [1, 1, 1, 1, 1, 1], [0, 0, 1, 1, 1, 1], [0, 1, 1, 1, 1, 1]])
    out = lm.generate(inputs_embeds=emb, attention_mask=mask, max_length=6 + 5)
    assert out.shape == (4, 5)
    calls = lm._engine.calls
    assert sorted((B, S) for B, S, _, _ in calls) == [(1, 5), (1, 6), (2, 4)]             # one call per real length
    assert all(budget == 5 for _, _, budget, _ in calls)                                  # budget counted from the PADDED length
    # every row equals what the same row alone, without its padding, produces
    for b in range(4):
        solo = _fake_lm().generate(inputs_embeds=emb[b:b + 1, mask[b].bool()], max_length=int(mask[b].sum()) + 5)
        assert torch.equal(solo[0], out[b])
    with pytest.raises(ValueError):
        lm.generate(inputs_embeds=emb, attention_mask=torch.tensor([[1, 1, 1, 1, 1, 0]] * 4), max_length=12)


def test_padded_prompts_row0_stop_ends_every_group():
    lm = _fake_lm()
    emb = torch.ones(3, 4, 2)
    emb[1] *= 2
    mask = torch.tensor([[0, 1, 1, 1], [1, 1, 1, 1], [0, 1, 1, 1]])
    free = lm.generate(inputs_embeds=emb, attention_mask=mask, max_length=4 + 8)
    stop = free[0, 2:4].tolist()                                                           # row 0 emits this pair at steps 2-3
    crit = [StoppingCriteriaSub(stops=[stop])]
    got = lm.generate(inputs_embeds=emb, attention_mask=mask, max_length=4 + 8, stopping_criteria=crit)
    assert got.shape == (3, 4) and torch.equal(got, free[:, :4])                           # all rows cut where row 0 stopped
    # only the group that contains row 0 is given the stop sequence
    with_stop = [c for c in lm._engine.calls[-2:] if c[3]]
    assert len(with_stop) == 1 and with_stop[0][0] == 2


def test_streamer_protocol_on_host():
    lm = _fake_lm()

    class S:
        def __init__(self):
            self.v, self.done = [], False

        def put(self, x):
            self.v.append(x.clone())

        def end(self):
            self.done = True

    st = S()
    out = lm.generate(inputs_embeds=torch.ones(2, 3, 2), max_length=3 + 6, streamer=st)
    assert st.done and st.v[0].shape == (2, 0) and torch.equal(torch.stack(st.v[1:], 1), out)
    with pytest.raises(ValueError):
        lm.generate(inputs_embeds=torch.ones(2, 3, 2), max_length=9, streamer=S(), num_beams=2)


def test_min_length_is_reduced_by_the_prompt_length():
    """HF _prepare_generated_length: with inputs_embeds, min_length -= prompt length (floored at 0)."""
    lm = _fake_lm()
    emb = torch.ones(1, 4, 2)
    lm.generate(inputs_embeds=emb, min_length=10, max_length=20)
    assert lm._engine.last_kw["min_new_tokens"] == 6
    lm.generate(inputs_embeds=emb, min_length=3, max_length=20)              # the im2svg situation: nothing left
    assert "min_new_tokens" not in lm._engine.last_kw
    lm.generate(inputs_embeds=emb, min_length=10, max_length=20, num_beams=2)     # beams: the scorer masks EOS in the log-probs
    assert lm._engine.last_kw["min_new_tokens"] == 6 and lm._engine.last_kw["num_beams"] == 2
    # padded rows: the PADDED length is what HF subtracts, so every length group gets the same number of EOS-free steps
    emb2 = torch.ones(2, 6, 2)
    mask = torch.tensor([[0, 0, 1, 1, 1, 1], [1, 1, 1, 1, 1, 1]])
    lm.generate(inputs_embeds=emb2, attention_mask=mask, min_length=9, max_length=6 + 5)
    assert lm._engine.last_kw["min_new_tokens"] == 3 and [c[1] for c in lm._engine.calls[-2:]] == [4, 6]


def test_token_callback_keeps_the_first_exception():
    """An exception of the streaming callback cannot unwind through sv_generate: it is kept and re-raised afterwards."""
    import ctypes as C
    from starvector_amd.engine import token_callback
    got = []
    cb, errs = token_callback(lambda t, c: got.append((t.clone(), c)))
    arr = (C.c_int32 * 6)(1, 2, 3, 4, 5, 6)
    cb(None, arr, 2, 7, 3)
    assert not errs and got[0][1] == 7 and got[0][0].tolist() == [[1, 2, 3], [4, 5, 6]] and got[0][0].dtype == torch.int64

    calls = []

    def bad(t, c):
        calls.append(c)
        raise RuntimeError("boom")

    cb2, errs2 = token_callback(bad)
    cb2(None, arr, 2, 0, 3)
    cb2(None, arr, 2, 3, 3)                                  # skipped: the stream is already broken
    assert calls == [0] and len(errs2) == 1 and isinstance(errs2[0], RuntimeError)


def test_config_from_checkpoint_reads_sizes_off_the_tensors():
    """Offline, the numbers the reference takes from the HF sub-model configs come from the checkpoint's own tensors."""
    from starvector_amd.model import config_from_checkpoint
    pd = "model.svg_transformer.transformer.transformer."
    c1 = config_from_checkpoint({"starcoder_model_name": "bigcode/starcoderbase-1b", "image_encoder_type": "clip",
                                 "hidden_size": 2048, "num_hidden_layers": 24, "num_attention_heads": 16, "vocab_size": 49152,
                                 "torch_dtype": "float16", "max_length": 8192},
                                {pd + "wte.weight": (49156, 2048), pd + "wpe.weight": (8192, 2048),
                                 pd + "h.0.mlp.c_fc.weight": (8192, 2048)})
    e1 = c1.engine_config()
    assert (e1.vocab, e1.n_positions, e1.n_inner, e1.arch) == (49156, 8192, 8192, "v1")
    p2 = "model.svg_transformer.transformer.model."
    c2 = config_from_checkpoint({"starcoder_model_name": "bigcode/starcoder2-7b", "image_encoder_type": "siglip_384",
                                 "hidden_size": 4608, "num_hidden_layers": 32, "num_attention_heads": 36, "num_kv_heads": 4,
                                 "vocab_size": 49152, "max_length": 16000},
                                {p2 + "embed_tokens.weight": (49157, 4608), p2 + "layers.0.mlp.c_fc.weight": (18432, 4608)})
    e2 = c2.engine_config()
    assert (e2.vocab, e2.n_positions, e2.n_inner, e2.arch, e2.max_seq_len) == (49157, 16384, 18432, "v2", 16000)
    # without tensors the arch-aware defaults apply; explicit entries always win
    assert StarVectorConfig(starcoder_model_name="bigcode/starcoder2-7b").added_tokens == 5
    assert StarVectorConfig().added_tokens == 4 and StarVectorConfig().n_positions == 8192
    c3 = config_from_checkpoint({"vocab_size": 49152, "added_tokens": 4, "n_inner": 1024}, {pd + "wte.weight": (49200, 2048)})
    assert c3.added_tokens == 4 and c3.n_inner == 1024
    with pytest.raises(ValueError):
        config_from_checkpoint({"vocab_size": 49152}, {pd + "wte.weight": (100, 2048)})


def test_scoring_forward_padded_masks():
    """starvector_arch.py:161-184 with the mask a GRPO trainer passes (completions padded after EOS)."""
    import types
    from starvector_amd.model import StarVectorForCausalLM

    class Eng:
        def __init__(self):
            self.seen = None

        def embed_tokens(self, ids):
            return ids.float().unsqueeze(-1).expand(-1, -1, 4).to(torch.bfloat16)

        def forward_logits(self, emb, keep):
            self.seen = (tuple(emb.shape), keep)
            self.calls.append((tuple(emb.shape), keep))
            # "logit" = the embedding value of the position it belongs to, so placement can be checked
            pos = emb[:, -(keep or emb.shape[1]):, 0].float()
            return pos.unsqueeze(-1).expand(-1, -1, 7).contiguous()

    m = StarVectorForCausalLM.__new__(StarVectorForCausalLM)
    torch.nn.Module.__init__(m)
    eng = Eng()
    eng.calls = []
    object.__setattr__(m, "engine", eng)
    object.__setattr__(m, "model", types.SimpleNamespace(_get_embeddings=eng.embed_tokens))
    vis = torch.ones(1, 3, 4, dtype=torch.bfloat16)
    ids = torch.tensor([[5, 6, 7, 0], [5, 6, 0, 0]])
    right = torch.tensor([[1, 1, 1, 1, 1, 1, 1], [1, 1, 1, 1, 1, 0, 0]])
    out = m.forward(vis, ids, 2, right, 4)
    assert out.logits.shape == (2, 4, 7) and eng.seen == ((2, 7, 4), 4) and len(eng.calls) == 1
    # left padding: the row is scored without its pads (HF: masked keys, positions cumsum(mask) - 1) in its own engine pass
    eng.calls.clear()
    out = m.forward(vis, ids, 2, torch.tensor([[1] * 7, [0, 0, 1, 1, 1, 1, 1]]), 4)
    assert sorted(eng.calls) == [((1, 5, 4), 4), ((1, 7, 4), 4)]
    assert out.logits.shape == (2, 4, 7)
    assert out.logits[0, :, 0].tolist() == [5, 6, 7, 0] and out.logits[1, :, 0].tolist() == [5, 6, 0, 0]
    eng.calls.clear()
    out = m.forward(vis, ids, 2, torch.tensor([[0, 1, 1, 1, 1, 1, 0], [0, 1, 1, 1, 1, 1, 1]]), None)    # all positions, left + right pads
    assert eng.calls == [((2, 6, 4), 0)] and out.logits.shape == (2, 7, 7)
    assert out.logits[0, :, 0].tolist() == [0, 1, 1, 5, 6, 7, 0]          # column 0 = the pad: left at zero
    with pytest.raises(ValueError):
        m.forward(vis, ids, 2, torch.tensor([[1] * 7, [0, 0, 0, 0, 1, 1, 1]]), 4)        # the kept logits reach into the pads
    with pytest.raises(NotImplementedError):
        m.forward(vis, ids, 2, torch.tensor([[1] * 7, [1, 1, 0, 1, 1, 1, 1]]), 4)        # a hole
    with pytest.raises(NotImplementedError):
        m.forward(vis, ids, 2, torch.tensor([[1] * 7, [0, 1, 0, 1, 1, 1, 1]]), 4)        # a hole behind a left pad
    with pytest.raises(ValueError):
        m.forward(vis, ids, 2, torch.ones(2, 4, dtype=torch.long)[:, :3] * torch.tensor([[1, 1, 0]]), 4)
    with pytest.raises(ValueError, match="all zeros"):
        m.forward(vis, ids, 2, torch.tensor([[1] * 7, [0] * 7]), 4)                      # a row with no real position


def test_simple_starvector_processor_is_hf_style():
    """starvector_arch.py:17-90 (what `starvector.model.processor` is for a v1 checkpoint; scripts/quickstart-hf.py)."""
    from PIL import Image
    from starvector_amd import SimpleStarVectorProcessor
    tok = ByteTokenizer(49152)
    proc = SimpleStarVectorProcessor(tok, size=224)
    g = torch.Generator().manual_seed(1)
    rgb = Image.fromarray((torch.rand(100, 160, 3, generator=g) * 255).to(torch.uint8).numpy(), "RGB")
    one = proc(rgb, return_tensors="pt")["pixel_values"]
    assert one.shape == (3, 224, 224) and torch.equal(one, ImageTrainProcessor(size=224)(rgb))
    assert proc(images=[rgb, rgb]).pixel_values.shape == (2, 3, 224, 224)
    # RGBA: the alpha band is dropped (img.convert("RGB")), NOT composited on white like ImageTrainProcessor does
    rgba = Image.new("RGBA", (224, 224), (10, 200, 30, 0))
    x = proc(rgba)["pixel_values"]
    dropped = (torch.tensor([10, 200, 30]) / 255.0 - torch.tensor(sva.model.CLIP_MEAN)) / torch.tensor(sva.model.CLIP_STD)
    torch.testing.assert_close(x[:, 5, 5], dropped, rtol=0, atol=1e-6)
    white = (1.0 - torch.tensor(sva.model.CLIP_MEAN)) / torch.tensor(sva.model.CLIP_STD)
    torch.testing.assert_close(ImageTrainProcessor(size=224)(rgba)[:, 5, 5], white, rtol=0, atol=1e-5)
    enc = proc(text=["ab", "abcd"], images=rgb)
    assert enc["input_ids"].shape == (2, 4) and enc["pixel_values"].shape == (3, 224, 224)
    with pytest.raises(ValueError):
        proc()


class _FakeSlotEngine(_FakeEngine):
    """_FakeEngine plus the continuous-batching entry points (cb_admit / cb_step / cb_poll / cb_read / cb_reset): every slot
    runs the same per-row 'model' with its own prompt, budget, EOS and stop sequence."""

    def __init__(self, max_batch=8):
        super().__init__()
        from starvector_amd.engine import EngineConfig
        self.cfg = EngineConfig(max_batch=max_batch)
        self.slots = {}
        self.admits = []

    def cb_reset(self):
        self.slots = {}

    def cb_admit(self, emb, reqs):
        B, S, _ = emb.shape
        self.admits.append((B, S))
        base = emb.float().sum(dim=(1, 2)).round().long().tolist()
        out = []
        for b, r in zip(base, reqs):
            s = min(set(range(self.cfg.max_batch)) - set(self.slots))
            stream = [(b + 3 * t) % 97 + 1 for t in range(r["max_new_tokens"])]
            self.slots[s] = dict(stream=stream, n=1, live=True, stop=r.get("stop_ids"), eos=r.get("eos_token_id", -1))
            self._finish(s)
            out.append(s)
        return out

    def _finish(self, s):
        sl = self.slots[s]
        got = sl["stream"][:sl["n"]]
        if sl["n"] >= len(sl["stream"]) or got[-1] == sl["eos"] or (sl["stop"] and got[-len(sl["stop"]):] == list(sl["stop"])):
            sl["live"] = False

    def cb_step(self, n):
        for _ in range(n):
            for s, sl in self.slots.items():
                if sl["live"]:
                    sl["n"] += 1
                    self._finish(s)
        return sum(1 for sl in self.slots.values() if sl["live"])

    def cb_poll(self):
        lv = [int(self.slots[s]["live"]) if s in self.slots else 0 for s in range(self.cfg.max_batch)]
        st = [self.slots[s]["n"] if s in self.slots else 0 for s in range(self.cfg.max_batch)]
        return lv, st

    def cb_read(self, s, start, n):
        return torch.tensor(self.slots[s]["stream"][start:start + n], dtype=torch.long)


def _slot_lm():
    lm = HipCausalLM.__new__(HipCausalLM)
    torch.nn.Module.__init__(lm)
    object.__setattr__(lm, "_engine", _FakeSlotEngine())
    lm.eos_token_id, lm.pad_token_id, lm.seed = 0, 99, 0
    return lm


def test_padded_prompts_run_as_slots_of_one_decode_loop():
    """With the continuous-batching entry points available, rows of different real length decode TOGETHER (one prompt pass per
    length group, then one loop); every row equals its unpadded solo run; HF's row-0 stop cuts every row; nothing is left live."""
    lm = _slot_lm()
    torch.manual_seed(0)
    emb = torch.randint(0, 5, (4, 6, 3)).float()
    mask = torch.tensor([[0, 0, 1, 1, 1, 1], [1, 1, 1, 1, 1, 1], [0, 0, 1, 1, 1, 1], [0, 1, 1, 1, 1, 1]])
    out = lm.generate(inputs_embeds=emb, attention_mask=mask, max_length=6 + 7)
    assert out.shape == (4, 7) and lm._engine.calls == []                                   # no classic generate call at all
    assert sorted(lm._engine.admits) == [(1, 5), (1, 6), (2, 4)]                            # prompt passes per real length
    for b in range(4):
        solo = _fake_lm().generate(inputs_embeds=emb[b:b + 1, mask[b].bool()], max_length=int(mask[b].sum()) + 7)
        assert torch.equal(solo[0], out[b])
    assert lm._engine.slots == {}                                                           # the slots were reset
    stop = out[0, 2:4].tolist()
    cut = lm.generate(inputs_embeds=emb, attention_mask=mask, max_length=6 + 7, stopping_criteria=[StoppingCriteriaSub(stops=[stop])])
    assert cut.shape == (4, 4) and torch.equal(cut, out[:, :4])                             # row 0's stop ends every row
    with pytest.raises(ValueError):
        lm.generate(inputs_embeds=emb, attention_mask=torch.tensor([[1, 1, 1, 1, 1, 0]] * 4), max_length=12)


def test_reference_format_checkpoint_directory_round_trip(tmp_path):
    """A checkpoint directory as the reference's `from_pretrained` reads it (scripts/quickstart.py:9, starvector_arch.py:96-145; HF
    `save_pretrained`: config.json + sharded safetensors + index, tensors under train/util.py:71's names) -> the mirror's config.  The
    numbers the reference takes from the HF sub-models it instantiates (vocabulary after resize_token_embeddings, positions, MLP
    width) come off the saved tensors.  The GPU half (load + generate) is tests/test_gpu_e2e.py::test_from_pretrained_*."""
    import json
    from safetensors import safe_open
    from oracle import starvector_oracle as O
    from starvector_amd.model import config_from_checkpoint, StarVectorForCausalLM
    from starvector_amd._lib import StarVectorHipError
    from tests.ckpt_util import write_reference_checkpoint
    cfg = O.OracleConfig.tiny()
    w = O.make_weights(cfg, seed=5)
    d = str(tmp_path / "ckpt")
    write_reference_checkpoint(d, cfg, w, n_shards=3, torch_dtype="float16")
    files = sorted(os.listdir(d))
    assert files == ["config.json", "model-00001-of-00003.safetensors", "model-00002-of-00003.safetensors",
                     "model-00003-of-00003.safetensors", "model.safetensors.index.json"]
    shapes = {}
    for fn in files:
        if fn.endswith(".safetensors"):
            with safe_open(os.path.join(d, fn), "pt") as f:
                for k in f.keys():
                    shapes[k] = tuple(f.get_slice(k).get_shape())
    assert set(shapes) == set(w) - {O.K_LMH}                              # the tied head is not saved
    assert set(json.load(open(os.path.join(d, "model.safetensors.index.json")))["weight_map"]) == set(shapes)
    c = config_from_checkpoint(json.load(open(os.path.join(d, "config.json"))), shapes)
    ec = c.engine_config()
    assert (ec.vocab, ec.n_positions, ec.n_inner, ec.hidden, ec.n_layer, ec.n_head) == (cfg.vocab, cfg.n_positions, cfg.n_inner,
                                                                                        cfg.hidden, cfg.n_layer, cfg.n_head)
    assert (ec.image_size, ec.patch_size, ec.vit_width, ec.vit_layers, ec.vit_heads, ec.arch) == (56, 14, 128, 2, 2, "v1")
    assert c.added_tokens == 4 and ec.max_batch == 4
    # hub names cannot be fetched here and say so; a directory without tokenizer files must not silently get the byte tokenizer
    with pytest.raises(FileNotFoundError, match="hub download"):
        StarVectorForCausalLM.from_pretrained("starvector/starvector-1b-im2svg")
    if not torch.cuda.is_available():
        with pytest.raises((StarVectorHipError, FileNotFoundError)):      # no GPU: the engine refuses (never a CPU model)
            StarVectorForCausalLM.from_pretrained(d)


def test_bench_self_launch_command_and_environment():
    """`python bench.py --gpus N` with no launcher re-executes itself under torch.distributed.run exactly as the driver's contract
    spells the command (one rank per GPU, rendezvous on 127.0.0.1), and the HSA IPC flag is set before torch is imported."""
    import importlib.util
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("_bench_mod", os.path.join(root, "bench.py"))
    src = open(os.path.join(root, "bench.py")).read()
    assert src.index('os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")') < src.index("import torch")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cmd = mod.self_launch_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], port=29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[3:9] == ["--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1", "--master-port", "29511"]
    assert cmd[9] == os.path.join(root, "bench.py") and cmd[10:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    free = mod.self_launch_command(2, [])
    assert 1024 < int(free[8]) < 65536
    # a launcher whose world size disagrees with --gpus is refused before any GPU work
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "must agree" in (r.stderr + r.stdout)
    # the product benchmark does not need test infrastructure to describe the model: oracle/ is imported inside cpu_baseline only
    body = src[src.index("def main():"):]
    assert "from oracle" not in body and "import oracle" not in body


def test_import_sets_dmabuf_ipc_for_multi_process_gpu_work():
    """VERDICT r04 item 8 + ADVICE r05: a rank of a multi-process job (WORLD_SIZE > 1: generate_im2svg_dp under torchrun, RCCL) gets
    HSA_ENABLE_IPC_MODE_LEGACY=0 from the package import, as a default that an explicit launcher value overrides; a plain single-process
    import leaves the caller's environment alone."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys; sys.path.insert(0, %r); os.environ.pop('HSA_ENABLE_IPC_MODE_LEGACY', None); os.environ['WORLD_SIZE'] = '2'; "
            "import starvector_amd; print(os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'))" % root)
    run = lambda c: subprocess.run([sys.executable, "-c", c], capture_output=True, text=True, check=True).stdout.strip()
    assert run(code) == "0"
    assert run(code.replace("os.environ.pop('HSA_ENABLE_IPC_MODE_LEGACY', None)", "os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] = '1'")) == "1"
    assert run(code.replace("os.environ['WORLD_SIZE'] = '2'", "os.environ.pop('WORLD_SIZE', None)")) == "None"
