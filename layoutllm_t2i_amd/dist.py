"""Multi-GPU mode: independent replicas, ONE weight broadcast, no per-step collectives.

The reference's inference path is strictly single-GPU (txt2img.py:535-536, train_rl.py:321); images
are independent units, so the path shards by prompt with zero exchange during denoising.
One process per GPU (``torch.distributed``; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in CPU
tests).  Rank 0 loads + packs the checkpoint, then every packed tensor travels in a single
``broadcast`` of one flat byte buffer (2.5 GB fp16 UNet: one large xGMI-friendly message instead of
1238 small ones); the buffer's layout is the C engine's weight table (gl_weight_at), so every rank hands the received
bytes straight to gl_load_weights and carves its tensor views out of them by the same table.
"""
from __future__ import annotations

import contextlib

from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from .arch import UNetConfig, build_plan
from .weights import PackedWeights

_ALIGN = 256


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Round-robin shard: rank r takes items r, r+world, ....  Ragged tails are fine."""
    return list(range(rank, n_items, world))


def flatten_packed(P: PackedWeights) -> Tuple[torch.Tensor, dict]:
    """The flat byte buffer of P (its layout is the C engine's weight table, weights.PackedWeights.to_flat) + the few
    host-side facts the receivers need."""
    if P.flat is None:
        P.to_flat()
    return P.flat, dict(total=int(P.flat.numel()), has_sd_conv=bool(P.has_sd_conv))


def unflatten_packed(flat: torch.Tensor, manifest: dict, cfg: UNetConfig, device) -> PackedWeights:
    assert manifest["total"] == flat.numel()
    return PackedWeights.from_flat(flat, cfg, device, manifest["has_sd_conv"])


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


_DT = {"torch.float16": torch.float16, "torch.float32": torch.float32, "torch.int32": torch.int32, "torch.uint8": torch.uint8}


def flatten_tensors(tensors: Dict[str, torch.Tensor], device) -> Tuple[torch.Tensor, list]:
    """name -> tensor dict as ONE 256-byte-aligned uint8 buffer + a manifest [(name, dtype, shape, offset, nbytes)]."""
    man, off = [], 0
    for k in sorted(tensors):
        t = tensors[k]
        nb = t.numel() * t.element_size()
        man.append((k, str(t.dtype), tuple(t.shape), off, nb))
        off += (nb + _ALIGN - 1) // _ALIGN * _ALIGN
    flat = torch.zeros(max(off, _ALIGN), dtype=torch.uint8, device=device)
    for k, _, _, o, nb in man:
        if nb:
            flat[o:o + nb].copy_(tensors[k].contiguous().view(-1).view(torch.uint8))
    return flat, man


def unflatten_tensors(flat: torch.Tensor, manifest: list) -> Dict[str, torch.Tensor]:
    """Views into ``flat`` (no copies)."""
    out = {}
    for k, dt, shape, o, nb in manifest:
        out[k] = flat[o:o + nb].view(_DT[dt]).view(tuple(shape))
    return out


def broadcast_bundle(P: Optional[PackedWeights], vae_w: Optional[Dict[str, torch.Tensor]], cfg: Optional[UNetConfig], device,
                     src: int = 0, extra: Optional[dict] = None, aux: Optional[Dict[str, Dict[str, torch.Tensor]]] = None,
                     want_aux: bool = False):
    """The job's ONE data-path collective: the packed UNet buffer (2.5 GB, the C engine's weight-table layout)
    AND the packed VAE-decoder tensors (0.1 GB) travel in a single ``broadcast`` of one flat byte buffer; the few host-side
    facts (offsets, configs, ``extra``) go ahead of it as a pickled object.  Rank ``src`` passes its objects, the others
    pass None.  ``aux``: further named tensor dicts (the HIP text tower's weights) appended to the same buffer.
    Returns (PackedWeights, vae tensor dict or None, extra) on every rank -- plus the aux dicts when ``want_aux``; tensors are
    views of the buffer."""
    rank = dist.get_rank()
    if rank == src:
        uflat, uman = flatten_packed(P)
        vflat, vman = (flatten_tensors(vae_w, uflat.device) if vae_w is not None else (None, None))
        usz = int(uflat.numel())
        parts = [uflat] + ([vflat] if vflat is not None else [])
        aman = {}
        for name in sorted(aux or {}):
            af, am = flatten_tensors(aux[name], uflat.device)
            aman[name] = (sum(int(p_.numel()) for p_ in parts), int(af.numel()), am)
            parts.append(af)
        total = sum(int(p_.numel()) for p_ in parts)
        head = dict(unet=uman, unet_bytes=usz, vae=vman, vae_bytes=int(vflat.numel()) if vflat is not None else 0, total=total, cfg=cfg,
                    extra=extra, aux=aman)
        if len(parts) > 1:
            flat = torch.empty(total, dtype=torch.uint8, device=uflat.device)
            o = 0
            for p_ in parts:
                flat[o:o + p_.numel()].copy_(p_)
                o += int(p_.numel())
        else:
            flat = uflat
        box = [head]
    else:
        flat, box = None, [None]
    # RCCL stages object collectives on the CURRENT device: make that this rank's GPU even if the caller never set it
    d = torch.device(device)
    with (torch.cuda.device(d) if d.type == "cuda" else contextlib.nullcontext()):
        dist.broadcast_object_list(box, src=src)
        head = box[0]
        if rank != src:
            flat = torch.empty(head["total"], dtype=torch.uint8, device=device)
        dist.broadcast(flat.view(torch.int64) if flat.numel() % 8 == 0 else flat, src=src)      # the one collective
    if rank == src:
        return (P, vae_w, extra, aux) if want_aux else (P, vae_w, extra)      # the sender keeps its own objects (the staging buffer is dropped)
    usz = head["unet_bytes"]
    Pb = unflatten_packed(flat[:usz], head["unet"], head["cfg"], device)
    vw = unflatten_tensors(flat[usz:usz + head["vae_bytes"]], head["vae"]) if head["vae"] is not None else None
    if not want_aux:
        return Pb, vw, head["extra"]
    return Pb, vw, head["extra"], {name: unflatten_tensors(flat[o:o + nb], man) for name, (o, nb, man) in head["aux"].items()}


def checksum(P: PackedWeights) -> int:
    """Order-independent integer checksum of all packed bytes (broadcast-then-compare tests)."""
    tot = 0
    for k in sorted(P.w):
        b = P.w[k].contiguous().view(-1).view(torch.uint8)
        tot = (tot * 1000003 + int(b.to(torch.int64).sum().item()) + b.numel()) % (2 ** 61 - 1)
    return tot
