"""One rank of the world_size-2 gloo test of dist.broadcast_bundle (tests/test_dist_gloo.py): the packed UNet buffer and a
VAE-like tensor dict travel in ONE broadcast; every rank reports checksums."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

from layoutllm_t2i_amd import recipe
from layoutllm_t2i_amd.arch import TINY
from layoutllm_t2i_amd.dist import broadcast_bundle, checksum
from layoutllm_t2i_amd.weights import pack_state_dict


def vae_like():
    g = torch.Generator().manual_seed(5)
    return {"decoder.conv_in.w": torch.randn(7, 3, 5, generator=g).half(), "decoder.conv_in.b": torch.randn(7, generator=g),
            "decoder.norm_out.g": torch.randn(13, generator=g), "empty": torch.zeros(0)}


def tsum(d):
    return sorted((k, str(v.dtype), list(v.shape), float(v.double().sum())) for k, v in d.items())


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = {"n": 0}
    orig = dist.broadcast

    def counting(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)
    dist.broadcast = counting
    try:
        P = vw = None
        if rank == 0:
            P = pack_state_dict(recipe.state_dict(TINY, 0), TINY, "cpu", recipe.sd_first_conv(TINY, 0))
            vw = vae_like()
        Pb, vb, extra = broadcast_bundle(P, vw, TINY if rank == 0 else None, "cpu", src=0, extra=dict(tag="hello", n=3) if rank == 0 else None)
        print("RESULT " + json.dumps(dict(rank=rank, checksum=checksum(Pb), has_sd=bool(Pb.has_sd_conv), vae=tsum(vb), extra=extra,
                                          tensor_broadcasts=calls["n"], cfg_ok=bool(Pb.cfg == TINY))), flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
