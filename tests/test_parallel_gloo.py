"""CPU, world_size 2, gloo: the data-parallel wrapper (shard by rank -> local generate -> ONE all_gather of token
streams) driven through the REAL mirror classes (`StarVectorStarCoder`: tokenizer, `_prepare_generation_inputs`,
`generate_im2svg_grpo`, `HipCausalLM.generate`) over a scripted engine, so no GPU is needed.

What is checked: every rank ends up with the whole batch in global order and it equals the single-process run; the
collective count is exactly one (`all_gather_into_tensor`; no all_reduce, no broadcast, no barrier); ragged shards and
ragged widths; `num_return_sequences`; per-rank random streams; lengths travel in column 0 of the gathered block."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build_model():
    """The real mirror classes over a scripted engine: the 'decoder' emits, for a row whose image mean is m, the bytes of
    f'<{m}>' repeated; rows with an odd m stop after 5 tokens less than the budget of the call (ragged widths)."""
    from starvector_amd.engine import EngineConfig
    from starvector_amd.model import ByteTokenizer, StarVectorConfig, StarVectorStarCoder

    tok = ByteTokenizer(49152)

    class Engine:
        device = 0

        def __init__(self):
            self.cfg = EngineConfig(image_size=28, patch_size=14, vit_width=4, hidden=8, vocab=len(tok))
            self.calls = []

        def encode_image(self, image):
            return image.float().mean(dim=(1, 2, 3)).view(-1, 1, 1).expand(-1, self.cfg.query_length, 4).to(torch.bfloat16)

        def adapter(self, h):
            return torch.cat([h, h], dim=-1)

        def embed_tokens(self, ids):
            return (ids.float().unsqueeze(-1) / 300.0).expand(-1, -1, 8).to(torch.bfloat16)

        def generate(self, inputs_embeds, max_length, pad_token_id=0, seed=0, do_sample=False, **kw):
            B, S, _ = inputs_embeds.shape
            budget = max_length - S
            self.calls.append(dict(B=B, S=S, budget=budget, seed=seed, do_sample=do_sample))
            ms = [int(round(float(inputs_embeds[b, 0, 0]))) for b in range(B)]
            n = max(budget - (5 if m % 2 else 0) for m in ms)
            out = torch.full((B, n), pad_token_id, dtype=torch.long)
            for b, m in enumerate(ms):
                script = tok.encode(f"<{m}>") * budget
                k = budget - (5 if m % 2 else 0)
                out[b, :k] = torch.tensor(script[:k])
                if do_sample:
                    out[b, 0] = 40 + (seed + b) % 50            # a 'random' first token: depends on (seed, local row)
            return out

    class Model:
        def __init__(self):
            self.config = StarVectorConfig()
            self.engine = Engine()
            self.model = StarVectorStarCoder(self.config, self.engine, tok)

        def generate_im2svg(self, batch, **kw):
            return self.model.generate_im2svg(batch, **kw)

    return Model()


def _images(n):
    return torch.arange(n, dtype=torch.float32).view(n, 1, 1, 1).expand(n, 3, 28, 28).contiguous()


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import starvector_amd.parallel as par

    counts = {}
    for name in ("all_gather_into_tensor", "all_reduce", "all_gather", "broadcast", "barrier", "reduce_scatter_tensor"):
        real = getattr(dist, name)

        def wrap(*a, __real=real, __name=name, **k):
            counts[__name] = counts.get(__name, 0) + 1
            return __real(*a, **k)
        setattr(dist, name, wrap)

    model = _build_model()
    S_max = model.model.query_length + 4 + 30          # 5 visual rows (28 / 14 squared + class token) + '<svg' + 30 new tokens
    batch = {"image": _images(n)}
    out = par.generate_im2svg_dp(model, batch, max_length=S_max, num_beams=1, use_nucleus_sampling=False)
    n_coll = dict(counts)
    many = par.generate_im2svg_dp(model, batch, max_length=S_max, num_return_sequences=2, use_nucleus_sampling=True)
    seeds = [c["seed"] for c in model.engine.calls if c["do_sample"]]
    # raw collective: ragged widths, ragged shard sizes, lengths in column 0
    lo, hi = par.shard_bounds(n, rank, world)
    local = torch.full((hi - lo, 2 + rank), 7 + rank, dtype=torch.int64)
    full, lens = par.all_gather_token_streams(local, 99, n, width=6, return_lengths=True)
    try:
        par.all_gather_token_streams(local, 99, n)          # no agreed width: refused, never a second collective
        refused = False
    except ValueError:
        refused = True
    q.put((rank, out, n_coll, many, seeds, full.tolist(), lens.tolist(), refused))
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [5, 8])
def test_dp_generate_matches_single_process(n):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    from starvector_amd.parallel import shard_bounds
    model = _build_model()
    S_max = model.model.query_length + 4 + 30
    expect = model.generate_im2svg({"image": _images(n)}, max_length=S_max, num_beams=1, use_nucleus_sampling=False)
    assert len(expect) == n and all(s.startswith("<svg") for s in expect) and expect[3].startswith("<svg<3>")
    seeds_by_rank = {}
    for rank, out, n_coll, many, seeds, full, lens, refused in res:
        assert out == expect                                   # every rank holds the whole batch, in global order
        assert n_coll == {"all_gather_into_tensor": 1}         # exactly one collective, nothing else
        assert len(many) == 2 * n                              # num_return_sequences rows per image, global order
        for i in range(n):
            assert many[2 * i][5:] == expect[i][5:] and many[2 * i + 1][5:] == expect[i][5:]   # same image -> same script
        assert refused
        assert len(full) == n and all(len(r) == 6 for r in full)
        for r in range(world):
            lo, hi = shard_bounds(n, r, world)
            for row, ln in zip(full[lo:hi], lens[lo:hi]):
                assert ln == 2 + r and row[: 2 + r] == [7 + r] * (2 + r) and all(x == 99 for x in row[2 + r:])
        seeds_by_rank[rank] = seeds
    assert seeds_by_rank[0] and seeds_by_rank[1] and seeds_by_rank[0][0] != seeds_by_rank[1][0]    # per-rank random streams


def test_world_of_one_is_a_passthrough():
    from starvector_amd.parallel import all_gather_token_streams
    t = torch.arange(6).view(2, 3)
    assert all_gather_token_streams(t, 0, 2) is t
