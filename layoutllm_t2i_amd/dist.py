"""Multi-GPU mode: independent replicas, ONE weight broadcast, no per-step collectives.

The reference's inference path is strictly single-GPU (txt2img.py:535-536, train_rl.py:321); images
are independent units, so the path shards by prompt with zero exchange during denoising.
One process per GPU (``torch.distributed``; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in CPU
tests).  Rank 0 loads + packs the checkpoint, then every packed tensor travels in a single
``broadcast`` of one flat byte buffer (2.5 GB fp16 UNet: one large xGMI-friendly message instead of
1238 small ones); the other ranks carve views out of it by a small manifest sent alongside.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from .arch import UNetConfig, build_plan
from .weights import PackedWeights

_ALIGN = 256


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Round-robin shard: rank r takes items r, r+world, ....  Ragged tails are fine."""
    return list(range(rank, n_items, world))


def _layout(entries: List[Tuple[str, Tuple[int, ...], torch.dtype]]) -> Tuple[Dict[str, int], int]:
    off, offsets = 0, {}
    for name, shape, dtype in entries:
        n = 1
        for s in shape:
            n *= s
        offsets[name] = off
        off += (n * torch.empty((), dtype=dtype).element_size() + _ALIGN - 1) // _ALIGN * _ALIGN
    return offsets, off


def flatten_packed(P: PackedWeights) -> Tuple[torch.Tensor, dict]:
    """All tensors of P into one uint8 buffer (256-byte aligned slots) + the manifest to rebuild them."""
    entries = [(k, tuple(v.shape), v.dtype) for k, v in P.w.items()]
    offsets, total = _layout(entries)
    flat = torch.zeros(total, dtype=torch.uint8, device=P.device)
    for name, shape, dtype in entries:
        src = P.w[name].contiguous().view(-1).view(torch.uint8)
        flat[offsets[name]:offsets[name] + src.numel()].copy_(src)
    manifest = dict(entries=entries, scalars=dict(P.s), emb_offsets=dict(P.emb_offsets), emb_total=P.emb_total, total=total)
    return flat, manifest


def unflatten_packed(flat: torch.Tensor, manifest: dict, cfg: UNetConfig, device) -> PackedWeights:
    P = PackedWeights(cfg, build_plan(cfg), device)
    offsets, total = _layout(manifest["entries"])
    assert total == manifest["total"] == flat.numel()
    for name, shape, dtype in manifest["entries"]:
        n = 1
        for s in shape:
            n *= s
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        P.w[name] = flat[offsets[name]:offsets[name] + nbytes].view(dtype).view(shape)
    P.s.update(manifest["scalars"])
    P.emb_offsets.update(manifest["emb_offsets"])
    P.emb_total = manifest["emb_total"]
    return P


def broadcast_packed(P: Optional[PackedWeights], cfg: UNetConfig, device, src: int = 0) -> PackedWeights:
    """Rank ``src`` passes its PackedWeights, the others pass None; returns a PackedWeights on every
    rank whose tensors are views of the (single) broadcast buffer."""
    rank = dist.get_rank()
    if rank == src:
        flat, manifest = flatten_packed(P)
        box = [manifest]
    else:
        flat, box = None, [None]
    dist.broadcast_object_list(box, src=src)
    manifest = box[0]
    if rank != src:
        flat = torch.empty(manifest["total"], dtype=torch.uint8, device=device)
    # the one data-path collective of the whole job; sent as int64 words (the buffer is 256-byte padded) so the element
    # count of the 2.5 GB buffer stays far below 2^31 whatever the backend's count type
    dist.broadcast(flat.view(torch.int64) if flat.numel() % 8 == 0 else flat, src=src)
    return unflatten_packed(flat, manifest, cfg, device)


def checksum(P: PackedWeights) -> int:
    """Order-independent integer checksum of all packed bytes (broadcast-then-compare tests)."""
    tot = 0
    for k in sorted(P.w):
        b = P.w[k].contiguous().view(-1).view(torch.uint8)
        tot = (tot * 1000003 + int(b.to(torch.int64).sum().item()) + b.numel()) % (2 ** 61 - 1)
    return tot
