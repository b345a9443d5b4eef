"""CPU, world_size 2, gloo: the data-parallel wrapper (shard by rank -> local generate -> ONE all_gather
of token streams), with a stub model so no GPU is needed."""
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


class _StubTok:
    pad_token_id = 99

    def batch_decode(self, ids, skip_special_tokens=True):
        return [" ".join(str(int(t)) for t in row if int(t) != self.pad_token_id) for row in ids]


class _StubInner:
    """generate_im2svg_grpo of a fake model: row i of the GLOBAL batch (identified by its image value)
    yields tokens [i, i, ...] of length 3 + (i % 3); the local width is the longest local row."""

    class _T:
        tokenizer = _StubTok()
    svg_transformer = _T()

    def generate_im2svg_grpo(self, batch, **kw):
        ids = batch["image"].flatten().tolist()
        n = max(3 + (int(i) % 3) for i in ids)
        out = torch.full((len(ids), n), 99, dtype=torch.int64)
        for r, i in enumerate(ids):
            out[r, : 3 + (int(i) % 3)] = int(i)
        return {"outputs": out}


class _StubModel:
    model = _StubInner()

    def generate_im2svg(self, batch, **kw):
        tok = self.model.svg_transformer.tokenizer
        return tok.batch_decode(self.model.generate_im2svg_grpo(batch)["outputs"])


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from starvector_amd.parallel import all_gather_token_streams, generate_im2svg_dp, shard_bounds
    batch = {"image": torch.arange(n).view(n, 1)}
    out = generate_im2svg_dp(_StubModel(), batch)
    # raw collective: ragged widths and ragged shard sizes
    lo, hi = shard_bounds(n, rank, world)
    local = torch.full((hi - lo, 2 + rank), 7 + rank, dtype=torch.int64)
    full = all_gather_token_streams(local, 99, n)
    q.put((rank, out, full.tolist()))
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
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = _StubModel().generate_im2svg({"image": torch.arange(n).view(n, 1)})
    from starvector_amd.parallel import shard_bounds
    for rank, out, full in res:
        assert out == expect                       # every rank holds the whole batch, in global order
        assert len(full) == n and all(len(r) == 3 for r in full)      # width = max over ranks
        for r in range(world):
            lo, hi = shard_bounds(n, r, world)
            for row in full[lo:hi]:
                assert row[: 2 + r] == [7 + r] * (2 + r) and all(x == 99 for x in row[2 + r:])


def test_world_of_one_is_a_passthrough():
    from starvector_amd.parallel import all_gather_token_streams
    t = torch.arange(6).view(2, 3)
    assert all_gather_token_streams(t, 0, 2) is t
