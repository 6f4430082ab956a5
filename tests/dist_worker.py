"""One rank of the world_size-2 gloo test (launched by tests/test_dist_gloo.py as a subprocess)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

from layoutllm_t2i_amd import recipe
from layoutllm_t2i_amd.arch import TINY
from layoutllm_t2i_amd.dist import broadcast_packed, checksum, shard_indices
from layoutllm_t2i_amd.weights import pack_state_dict


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P = None
        if rank == 0:
            P = pack_state_dict(recipe.state_dict(TINY, 0), TINY, "cpu", recipe.sd_first_conv(TINY, 0))
        Pb = broadcast_packed(P, TINY, "cpu", src=0)
        mine = shard_indices(7, rank, world)
        summary = dict(rank=rank, checksum=checksum(Pb), n=len(Pb.w), emb_total=Pb.emb_total, scalars=sorted(Pb.s.items())[:3],
                       shard=mine, d16=str(Pb.w["emb_all.w"].dtype), d32=str(Pb.w["out.0.g"].dtype),
                       local_checksum=(checksum(P) if P is not None else None))
        print("RESULT " + json.dumps(summary), flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
