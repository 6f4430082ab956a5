"""One rank of the product-level multi-GPU test (tests/test_gpu_dist.py): two ranks over gloo share ONE GPU, rank 0 reads the
synthetic checkpoint, load_all_models_sharded moves UNet + VAE in one broadcast, generate_batch_images_sharded shards the
prompts round-robin and gathers the images on rank 0."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np
import torch
import torch.distributed as dist

import stubs
from layoutllm_t2i_amd import interface as itf

PROMPTS = ["cat sitting on mat and dog under a tree", "a quiet empty street", "two birds on a wire", "red car near a house", "a boat"]
PHRASES = [["cat", "mat", "a big dog"], ["street"], ["bird", "bird"], ["red car", "house"], ["boat"]]
BOXES = [[[0.10, 0.10, 0.50, 0.55], [0.05, 0.60, 0.95, 0.95], [0.55, 0.20, 0.90, 0.70]], [[0.0, 0.5, 1.0, 1.0]],
         [[0.1, 0.2, 0.3, 0.4], [0.6, 0.2, 0.8, 0.4]], [[0.1, 0.5, 0.5, 0.9], [0.5, 0.1, 0.95, 0.8]], [[0.2, 0.4, 0.8, 0.9]]]
SEEDS = [11, 22, 33, 44, 55]
STEPS, LATENT = 4, 16


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ckpt, out_dir = sys.argv[1], sys.argv[2]
    dev = "cuda:0"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    reads = {"n": 0}
    orig_load = torch.load

    def counting_load(*a, **k):
        if a and isinstance(a[0], (str, os.PathLike)):      # file reads only (object collectives unpickle tensors through torch.load too)
            reads["n"] += 1
        return orig_load(*a, **k)
    torch.load = counting_load
    try:
        stubs.install_fake_sng_parser()
        strict = os.environ.get("W_STRICT") == "1"          # rank 0 decides (its argument); the flag and the split layout travel in the bundle
        am = itf.load_all_models_sharded(ckpt, dev, src=0, strict=strict if rank == 0 else None)
        if os.environ.get("W_CLIP_TOWER"):
            # checkpoint with a CLIP text tower: the HIP encoder is on every rank (its weights came in the bundle), rank 0 only tokenises;
            # grounding phrases go through the same tower (clip_model None on every rank)
            clip, proc = None, (stubs.ToyProcessor() if rank == 0 else None)
        else:
            clip, proc = (stubs.toy_clip().to(dev), stubs.ToyProcessor()) if rank == 0 else (None, None)
        args = (PROMPTS, PHRASES, BOXES) if rank == 0 else (None, None, None)
        imgs = itf.generate_batch_images_sharded(am, *args, clip, proc, device=dev, seeds=SEEDS if rank == 0 else None, src=0,
                                                 steps=STEPS, latent=LATENT)
        if rank == 0:
            np.savez(os.path.join(out_dir, "images.npz"), imgs=np.stack([np.asarray(im) for im in imgs]))
        print("RESULT " + json.dumps(dict(rank=rank, ckpt_reads=reads["n"], text_encoder=am[2] is not None, text_encoder_type=type(am[2]).__name__,
                                          n_images=None if imgs is None else len(imgs), strict=bool(getattr(am[0], "strict", False)),
                                          split_weights=bool(getattr(am[0].cfg, "split_weights", False)))), flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
