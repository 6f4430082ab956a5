"""Build-container experiment (CPU only): how much of the whole-UNet error of an fp16-storage pipeline comes from
rounding the RESIDUAL STREAM to fp16 at every add, and what floor remains when only the matmul / conv operands are
fp16 (fp32 accumulate, fp32 residual stream).  Runs the oracle under a TorchFunctionMode that injects roundings.

    python tools/precision_sim.py [tiny|full32|full64]
"""
import os
import sys
import time

import torch
import torch.nn.functional as F
from torch.overrides import TorchFunctionMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from layoutllm_t2i_amd import recipe
from layoutllm_t2i_amd.arch import TINY, UNetConfig
from layoutllm_t2i_amd.weights import random_state_dict
from oracle import unet_ref

h = lambda t: t.half().float() if torch.is_tensor(t) and t.dtype == torch.float32 else t


class Sim(TorchFunctionMode):
    def __init__(self, round_adds: bool, round_outputs: bool):
        super().__init__()
        self.round_adds, self.round_outputs = round_adds, round_outputs

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in (F.linear, F.conv2d):
            a = list(args)
            a[0], a[1] = h(a[0]), h(a[1])
            out = func(*a, **kwargs)
            return h(out) if self.round_outputs else out
        if func is torch.matmul:
            out = func(h(args[0]), h(args[1]))
            return h(out) if self.round_outputs else out
        if func in (F.layer_norm, F.group_norm):
            out = func(*args, **kwargs)
            return h(out)
        if func in (torch.Tensor.add, torch.Tensor.__add__, torch.add, torch.Tensor.__radd__):
            out = func(*args, **kwargs)
            if self.round_adds and torch.is_tensor(args[0]) and torch.is_tensor(args[1]) and args[0].dim() >= 3 and args[1].dim() >= 3 \
                    and args[0].shape == args[1].shape:
                return h(out)
            return out
        return func(*args, **kwargs)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "tiny"
    torch.set_num_threads(8)
    if which == "tiny":
        cfg, hw = TINY, 16
        import numpy as np
        sd = {k: torch.from_numpy(np.asarray(v)).float() for k, v in recipe.state_dict(cfg, 0).items()}
    else:
        cfg, hw = UNetConfig(), int(which[4:])
        sd = random_state_dict(cfg, torch.device("cpu"), seed=3)
    sd = {k: (v.half().float() if v.dim() >= 2 else v) for k, v in sd.items()}
    inp = {k: torch.from_numpy(v) for k, v in recipe.synth_inputs(cfg, 1, hw, n_boxes=8, n_rel=3, seed=4321).items()}
    t = torch.full((1,), 481, dtype=torch.long)
    args = (sd, cfg, inp["x"].half().float(), t, inp["context"].half().float(), inp["relations"].half().float(), inp["boxes"], inp["masks"],
            inp["positive_embeddings"])
    with torch.no_grad():
        t0 = time.time()
        ref = unet_ref.unet_forward(*args)
        print(f"ref {time.time() - t0:.1f}s |ref|max={float(ref.abs().max()):.3f}")
        for name, ra, ro in (("operands only (fp32 stream, fp32 branch outputs)", False, False),
                             ("+ branch outputs fp16", False, True),
                             ("+ residual adds fp16 (round-1 engine)", True, True)):
            with Sim(ra, ro):
                out = unet_ref.unet_forward(*args)
            d = out - ref
            tol = 1e-4 + 1e-3 * ref.abs()
            print(f"{name:50s} rel_l2={float(d.norm() / ref.norm()):.3e} max|err|={float(d.abs().max()):.3e} "
                  f"outside rtol1e-3/atol1e-4: {float((d.abs() > tol).float().mean()) * 100:.1f}%")


if __name__ == "__main__":
    main()
