"""Data-parallel im2svg: one process per GPU, the batch is sharded by rank, every rank runs the whole
path on its shard (full weight replica, own paged KV pool), and EXACTLY ONE collective -- an
all_gather of the decoded token streams -- returns the result to every rank.
No collective inside the decode loop, no host synchronisation around the collective.

The reference never shards inference (one process, validation/starvector_hf_validator.py:56-58); what is
mirrored here is its `generate_im2svg` contract.  The padded width of a stream is known before anything
runs -- `max_length` counts the prompt rows (257 visual + P prompt ids), so a returned row [prompt ids | new
tokens] is at most `max_length - query_length` wide -- hence one fixed-shape gather suffices: each rank
contributes an int32 block [B_max_local, 1 + width] whose column 0 carries the row's length (0 for the padding
rows of a ragged shard) and whose remaining columns are the ids padded with pad_token_id.

torch.distributed backend "nccl" is RCCL on ROCm (xGMI inside a node); the gloo backend runs the same
code on CPU tensors for the world_size-2 tests.

The reference's row-0 `</svg>` stop (starvector_base.py:9-20) looks at row 0 of the batch a `generate` call sees:
under DP that is row 0 of every SHARD (there is no global row 0 inside a rank), so a shard may end earlier or later
than the single-process batch would; rows are never cut short of their own stop.  Stated in DESIGN.md section 6.
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
    out = {}
    for k, v in batch.items():
        if torch.is_tensor(v) and v.shape[:1] == (n,):
            out[k] = v[lo:hi]
        elif isinstance(v, (list, tuple)) and len(v) == n:
            out[k] = v[lo:hi]
        else:
            out[k] = v
    return out


def all_gather_token_streams(local: torch.Tensor, pad_token_id: int, global_batch: int, width: Optional[int] = None,
                             group: Optional[dist.ProcessGroup] = None, rows_per_item: int = 1,
                             return_lengths: bool = False):
    """local: int64 [B_local * rows_per_item, N_local] -> int64 [global_batch * rows_per_item, width], rows in global
    order, padded with pad_token_id (and, with return_lengths, the int64 [rows] number of columns each rank produced).

    `width` is the padded width every rank agrees on WITHOUT talking: max_length - query_length for
    [prompt ids | new tokens] streams.  It is required when world_size > 1 (N_local differs between ranks when a shard
    stops early).  One `all_gather_into_tensor`, no other collective, no `.item()`.
    """
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        if return_lengths:
            return local, torch.full((local.shape[0],), local.shape[1], dtype=torch.int64, device=local.device)
        return local
    if width is None:
        raise ValueError("all_gather_token_streams: `width` (the common padded width, e.g. max_length - query_length) is "
                         "required with more than one rank")
    world = dist.get_world_size(group)
    if local.shape[1] > width:
        raise ValueError(f"local streams are {local.shape[1]} wide, more than the agreed width {width}")
    sizes = [shard_bounds(global_batch, r, world) for r in range(world)]
    b_max = max(hi - lo for lo, hi in sizes) * rows_per_item
    if local.shape[0] > b_max:
        raise ValueError(f"{local.shape[0]} local rows exceed the largest shard ({b_max} rows)")
    dev = local.device
    block = torch.full((b_max, 1 + width), pad_token_id, dtype=torch.int32, device=dev)
    block[:, 0] = 0
    block[: local.shape[0], 0] = local.shape[1]
    block[: local.shape[0], 1: 1 + local.shape[1]] = local.to(torch.int32)
    out = torch.empty((world * b_max, 1 + width), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(out, block, group=group)
    rows = [out[r * b_max: r * b_max + (hi - lo) * rows_per_item] for r, (lo, hi) in enumerate(sizes)]
    full = torch.cat(rows, dim=0).to(torch.int64)
    if return_lengths:
        return full[:, 1:], full[:, 0]
    return full[:, 1:]


def generate_im2svg_dp(model, batch: dict, group: Optional[dist.ProcessGroup] = None, **kwargs) -> List[str]:
    """Drop-in for ``model.generate_im2svg(batch, **kw)`` under torchrun: every rank passes the SAME
    global batch, gets back the decoded SVG strings of the whole batch (num_return_sequences strings per image)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return model.generate_im2svg(batch, **kwargs)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = batch["image"].shape[0]
    local = shard_batch(batch, rank, world)
    inner = model.model
    tok = inner.svg_transformer.tokenizer
    nrs = int(kwargs.get("num_return_sequences", 1) or 1)
    # [prompt ids | new tokens]: max_length counts query_length visual rows + the prompt ids (starvector_base.py:228-241)
    width = int(kwargs.get("max_length", 30)) - int(inner.query_length)
    if width < 1:
        raise ValueError(f"max_length ({kwargs.get('max_length', 30)}) must exceed the {inner.query_length} visual rows")
    if "seed" not in kwargs and kwargs.get("use_nucleus_sampling", True):
        # every rank draws its per-call seed from its own torch generator (model.py HipCausalLM.generate); under torchrun the
        # generators start equal, so the rank is mixed in: local row i of different shards must not share a random stream
        kwargs = dict(kwargs, seed=(int(torch.randint(0, 2 ** 62, ()).item()) ^ (0x9E3779B97F4A7C15 * (rank + 1))) & (2 ** 63 - 1))
    if local["image"].shape[0] > 0:
        outputs = inner.generate_im2svg_grpo(local, **kwargs)["outputs"]
    else:
        outputs = torch.empty((0, 1), dtype=torch.int64, device=batch["image"].device)
    full = all_gather_token_streams(outputs, tok.pad_token_id, n, width, group, rows_per_item=nrs)
    return tok.batch_decode(full, skip_special_tokens=True)
