"""One rank of the multi-GPU correctness test (tests/test_gpu_dist.py): ranks share ONE GPU over gloo, receive the
packed weights by the single broadcast, take their round-robin shard of a fixed prompt list and denoise it.  Every
sample's starting noise and conditioning are a function of its PROMPT INDEX (not of the rank), so a sample's result
must not depend on which rank ran it (SURVEY 8e / section 4 multi-GPU row)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

from layoutllm_t2i_amd import recipe
from layoutllm_t2i_amd.arch import TINY
from layoutllm_t2i_amd.dist import broadcast_packed, checksum, shard_indices
from layoutllm_t2i_amd.engine import UNetEngine
from layoutllm_t2i_amd.interface import denoise
from layoutllm_t2i_amd.model import GroundingNetInput, LatentDiffusion, UNetModel
from layoutllm_t2i_amd.weights import pack_state_dict

N_PROMPTS, HW, STEPS = 6, 16, 6


def sample_inputs(idx):
    """conditioning + noise of prompt `idx`: seeded by the prompt index only"""
    return {k: torch.from_numpy(v) for k, v in recipe.synth_inputs(TINY, 1, HW, n_boxes=2 + idx % 3, n_rel=1 + idx % 3, seed=1000 + idx).items()}


def run_shard(model, indices, dev):
    parts = [sample_inputs(i) for i in indices]
    inp = {k: torch.cat([p[k] for p in parts], 0) for k in parts[0]}
    model.first_conv_type = "GLIGEN"
    batch = dict(boxes=inp["boxes"], masks=inp["masks"], text_embeddings=inp["positive_embeddings"])
    am = (model, None, None, LatentDiffusion(device=dev), {})
    return denoise(am, inp["context"], inp["uc"], inp["relations"], batch, inp["x"].to(dev), [0.5, 0.0, 0.5], 7.5, steps=STEPS).cpu()


def model_from_packed(P, dev):
    m = UNetModel.__new__(UNetModel)
    m.cfg, m.device = TINY, torch.device(dev)
    m.image_size, m.in_channels, m.out_channels, m.model_channels = TINY.image_size, TINY.in_channels, TINY.out_channels, TINY.model_channels
    m.first_conv_restorable, m.first_conv_type, m.allow_missing_sd_conv = True, "GLIGEN", False
    m.grounding_tokenizer_input = GroundingNetInput()
    m.fuser_scale, m.training, m._cond_key = 1.0, False, None
    m.engine = UNetEngine(P)
    return m


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out_dir = sys.argv[1]
    dev = "cuda:0"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P = None
        if rank == 0:
            P = pack_state_dict(recipe.state_dict(TINY, 0), TINY, torch.device(dev), recipe.sd_first_conv(TINY, 0))
        Pb = broadcast_packed(P, TINY, torch.device(dev), src=0)
        model = model_from_packed(Pb, dev)
        mine = shard_indices(N_PROMPTS, rank, world)
        lat = run_shard(model, mine, dev)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), idx=np.asarray(mine), lat=lat.numpy())
        print("RESULT " + json.dumps(dict(rank=rank, shard=mine, checksum=checksum(Pb), has_sd=bool(Pb.has_sd_conv))), flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
