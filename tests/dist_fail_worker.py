"""One rank of the world_size-2 gloo failure-propagation test (tests/test_dist_gloo.py): generate_batch_images_sharded must
raise on EVERY rank -- never leave a rank blocked in a collective -- when (a) src fails before the conditioning broadcast,
(b) one rank fails after it (encoding / denoising its shard)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

from layoutllm_t2i_amd import interface as itf


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    mode = sys.argv[1]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    got = None
    try:
        am = (object(), None, None, None, {})              # reference-style (non-HIP) text encoder slot: src would encode every row
        if mode == "src_fails":
            # captions=None on src: len(None) raises inside the src-only preparation
            args = (None, None, None)
        else:
            args = (["a", "b", "c"], [["x"]] * 3, [[[0.1, 0.1, 0.5, 0.5]]] * 3) if rank == 0 else (None, None, None)
            rows = dict(context=torch.zeros(3, 2, 4), uc=torch.zeros(3, 2, 4), relations=torch.zeros(3, 2, 4), boxes=torch.zeros(3, 30, 4),
                        masks=torch.zeros(3, 30), text_embeddings=torch.zeros(3, 30, 4))
            itf.prepare_conditioning = lambda *a, **k: rows          # src's preparation succeeds ...

            def shard(all_models, cond, noise, device, **kw):          # ... and rank 1's shard fails
                if rank == 1:
                    raise ValueError("boom on rank 1")
                import numpy as np
                return np.zeros((noise.shape[0], 8, 8, 3), dtype=np.uint8)
            itf.run_shard = shard
        try:
            itf.generate_batch_images_sharded(am, *args, None, None, device="cpu", seeds=None, src=0, steps=2, latent=8)
            got = "returned"
        except RuntimeError as ex:
            got = "RuntimeError: " + str(ex)[:200]
        print("RESULT " + json.dumps(dict(rank=rank, got=got)), flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
