"""Data-parallel im2svg: one process per GPU, the batch is sharded by rank, every rank runs the whole
path on its shard (full weight replica, own paged KV pool), and ONE all_gather at the end returns the
decoded token streams to every rank.  No collective inside the decode loop.

torch.distributed backend "nccl" is RCCL on ROCm (xGMI inside a node); the gloo backend runs the same
code on CPU tensors for the world_size-2 tests.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split of n items over world ranks (first n % world ranks get one extra)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(batch: dict, rank: int, world: int) -> dict:
    n = batch["image"].shape[0]
    lo, hi = shard_bounds(n, rank, world)
    return {k: (v[lo:hi] if torch.is_tensor(v) and v.shape[:1] == (n,) else v) for k, v in batch.items()}


def all_gather_token_streams(local: torch.Tensor, pad_token_id: int, global_batch: int,
                             group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """local: int64 [B_local, N_local] -> int64 [global_batch, N_max], rows in global order.

    Ranks may stop at different lengths (the row-0 stop is per shard), so each rank first learns the
    common width from a tiny all_reduce(max), pads with pad_token_id, and then contributes exactly one
    all_gather of a fixed-shape int32 block [B_max_local, N_max] (ragged shard sizes are padded too).
    """
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = local.device
    meta = torch.tensor([local.shape[1]], dtype=torch.int32, device=dev)
    dist.all_reduce(meta, op=dist.ReduceOp.MAX, group=group)
    n_max = int(meta.item())
    b_max = max(shard_bounds(global_batch, r, world)[1] - shard_bounds(global_batch, r, world)[0] for r in range(world))
    block = torch.full((b_max, n_max), pad_token_id, dtype=torch.int32, device=dev)
    block[: local.shape[0], : local.shape[1]] = local.to(torch.int32)
    out = torch.empty((world * b_max, n_max), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(out, block, group=group)
    rows = []
    for r in range(world):
        lo, hi = shard_bounds(global_batch, r, world)
        rows.append(out[r * b_max: r * b_max + (hi - lo)])
    return torch.cat(rows, dim=0).to(torch.int64)


def generate_im2svg_dp(model, batch: dict, group: Optional[dist.ProcessGroup] = None, **kwargs) -> List[str]:
    """Drop-in for ``model.generate_im2svg(batch, **kw)`` under torchrun: every rank passes the SAME
    global batch, gets back the decoded SVG strings of the whole batch."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return model.generate_im2svg(batch, **kwargs)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = batch["image"].shape[0]
    local = shard_batch(batch, rank, world)
    inner = model.model
    tok = inner.svg_transformer.tokenizer
    if local["image"].shape[0] > 0:
        outputs = inner.generate_im2svg_grpo(local, **kwargs)["outputs"]
    else:
        outputs = torch.empty((0, 1), dtype=torch.int64, device=batch["image"].device)
    full = all_gather_token_streams(outputs, tok.pad_token_id, n, group)
    return tok.batch_decode(full, skip_special_tokens=True)
