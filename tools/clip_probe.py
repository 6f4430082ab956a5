#!/usr/bin/env python3
"""Times the CLIP ViT-L/14 towers (random init) at the rollout's sizes on the GPU box; run under rocprofv3 --kernel-trace --stats
for the per-kernel split."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from layoutllm_t2i_amd.clip import ClipTowers

dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(1)
tw = ClipTowers(bench.random_clip_vit_l14_state_dict(dev, g), device=dev)
px = torch.randn(32, 3, 224, 224, device=dev, generator=g)
ids = torch.randint(1, 49406, (16, 77), device=dev, generator=g)
ids[:, 40] = 49407
for name, fn in (("vision 32 x 224^2", lambda: tw.get_image_features(px)), ("text 16 x 77", lambda: tw.get_text_features(ids))):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.time() - t0) / 5 * 1e3:.2f} ms", flush=True)
