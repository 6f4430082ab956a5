"""Experiment: does splitting the 2B CFG batch into two concurrently replayed half-batch graphs (two streams) beat
one 2B graph?  Timing only (the shared split-K workspace makes the concurrent results invalid)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from layoutllm_t2i_amd import recipe
from layoutllm_t2i_amd.arch import UNetConfig
from layoutllm_t2i_amd.engine import UNetEngine
from layoutllm_t2i_amd.model import GroundingNetInput
from layoutllm_t2i_amd.weights import pack_state_dict, random_state_dict

dev = torch.device("cuda:0")
cfg = UNetConfig()
sd = random_state_dict(cfg, dev, seed=0)
packed = pack_state_dict(sd, cfg, dev, None)
del sd


def make(Bn):
    inp = {k: torch.from_numpy(v).to(dev) for k, v in recipe.synth_inputs(cfg, Bn, 64, n_boxes=8, n_rel=3, seed=1).items()}
    e = UNetEngine(packed)
    e.set_conditioning(inp["context"], inp["relations"], inp["boxes"], inp["masks"], inp["positive_embeddings"], 64)
    x = torch.randn(Bn, 4, 64, 64, device=dev)
    return e, x


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


for scale in (1.0, 0.0):
    e8, x8 = make(8)
    t8 = timeit(lambda: e8.forward(x8, 500.0, scale))
    ea, xa = make(4)
    eb, xb = make(4)
    t4 = timeit(lambda: ea.forward(xa, 500.0, scale))
    eb.forward(xb, 500.0, scale)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def both():
        with torch.cuda.stream(s1):
            ea.forward(xa, 500.0, scale)
        with torch.cuda.stream(s2):
            eb.forward(xb, 500.0, scale)
    tb = timeit(both)
    print(f"fuser_scale {scale}: one Bn=8 graph {t8:.2f} ms | one Bn=4 graph {t4:.2f} ms | two Bn=4 graphs on two streams {tb:.2f} ms")
