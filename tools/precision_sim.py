"""Build-container experiment (CPU only): how much of the whole-UNet error of an fp16-storage pipeline comes from
rounding the RESIDUAL STREAM to fp16 at every add, and what floor remains when only the matmul / conv operands are
fp16 (fp32 accumulate, fp32 residual stream).  Runs the oracle under a TorchFunctionMode that injects roundings.

    python tools/precision_sim.py [tiny|full32|full64]
    python tools/precision_sim.py attribute [tiny|full32|full64]     per-op-class attribution of the fp16-operand floor:
        operands (activations in, weights) are rounded to fp16 for ONE class of matrix products at a time -- 3x3 convs,
        the three kinds of 1x1 conv (proj_in, proj_out, ResBlock skip), q|k|v projections, attention to_out, GEGLU projection, FF output projection,
        Q.K^T, P.V (P and V rounded), small conditioning-side Linears -- everything else exact fp32; rel-L2^2 of the
        classes adds up to the all-classes floor when the contributions are independent.
    python tools/precision_sim.py strict [tiny|full32|full64]        what must be split for north_star's tolerance (round 5, before the strict
        mode was built): classes LEFT on single-fp16 operands, everything else exact (= split: hi + lo carries ~22 bits) -- which subsets keep
        <= 1 % of the output elements outside rtol 1e-3 / atol 1e-4; the relation chain alone; and the fp32 oracle against an fp64 evaluation of
        itself (the floor any fp32 implementation sits on).
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


CLASSES = ("conv3x3", "out_conv", "in_conv", "proj_in", "proj_out", "skip1x1", "qkv", "attn_out", "ff_in", "ff_out", "qk", "pv", "cond")


def classify(name: str, w: torch.Tensor) -> str:
    if w.dim() == 4:
        if w.shape[-1] == 3:
            if name.startswith("out."):
                return "out_conv"          # the last conv writes eps itself
            if name.startswith("input_blocks.0.0."):
                return "in_conv"
            return "conv3x3"
        return "proj_in" if ".proj_in." in name else "proj_out" if ".proj_out." in name else "skip1x1"
    if ".to_q." in name or ".to_k." in name or ".to_v." in name:
        return "qkv"
    if ".to_out." in name:
        return "attn_out"
    if ".net.0.proj." in name:
        return "ff_in"
    if ".net.2." in name:
        return "ff_out"
    return "cond"          # time_embed, emb_layers, position_net, fuser.linear


class ClassSim(TorchFunctionMode):
    """fp16 operand rounding for the enabled classes only (fp32 accumulate, fp32 everything else)."""

    def __init__(self, wclass: dict, enabled):
        super().__init__()
        self.wclass, self.enabled, self.mm = wclass, set(enabled), 0

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in (F.linear, F.conv2d):
            if self.wclass.get(id(args[1]), "cond") in self.enabled:
                a = list(args)
                a[0], a[1] = h(a[0]), h(a[1])
                return func(*a, **kwargs)
            return func(*args, **kwargs)
        if func is torch.matmul:
            cls = "qk" if self.mm % 2 == 0 else "pv"      # oracle.attention: Q.K^T then P.V, strictly alternating
            self.mm += 1
            if cls in self.enabled:
                return func(h(args[0]), h(args[1]))
            return func(*args, **kwargs)
        return func(*args, **kwargs)


def attribute(which):
    torch.set_num_threads(8)
    if which == "tiny":
        import numpy as np
        cfg, hw = TINY, 16
        sd = {k: torch.from_numpy(np.asarray(v)).float() for k, v in recipe.state_dict(cfg, 0).items()}
    else:
        cfg, hw = UNetConfig(), int(which[4:])
        sd = random_state_dict(cfg, torch.device("cpu"), seed=3)
    sd = {k: (v.half().float() if v.dim() >= 2 else v) for k, v in sd.items()}
    wclass = {id(v): classify(k, v) for k, v in sd.items() if k.endswith(".weight") and v.dim() >= 2}
    inp = {k: torch.from_numpy(v) for k, v in recipe.synth_inputs(cfg, 1, hw, n_boxes=8, n_rel=3, seed=4321).items()}
    t = torch.full((1,), 481, dtype=torch.long)
    args = (sd, cfg, inp["x"].half().float(), t, inp["context"].half().float(), inp["relations"].half().float(), inp["boxes"], inp["masks"],
            inp["positive_embeddings"])
    with torch.no_grad():
        ref = unet_ref.unet_forward(*args)
        tol = 1e-4 + 1e-3 * ref.abs()
        rows = []
        for name, en in [("ALL classes", CLASSES)] + [(c, (c,)) for c in CLASSES] + [("all but proj_out", [c for c in CLASSES if c != "proj_out"]),
                                                                                     ("all but 1x1 convs", [c for c in CLASSES if c not in ("proj_in", "proj_out", "skip1x1")])]:
            t0 = time.time()
            with ClassSim(wclass, en):
                out = unet_ref.unet_forward(*args)
            d = out - ref
            r = float(d.norm() / ref.norm())
            rows.append((name, r))
            print(f"{name:18s} rel_l2={r:.3e}  rel_l2^2 share={r * r / (rows[0][1] ** 2) * 100:5.1f}%  max|err|={float(d.abs().max()):.3e} "
                  f"outside rtol1e-3/atol1e-4: {float((d.abs() > tol).float().mean()) * 100:5.1f}%   ({time.time() - t0:.0f}s)", flush=True)
        s2 = sum(r * r for n, r in rows[1:1 + len(CLASSES)])
        print(f"sum of single-class rel_l2^2 = {s2:.3e} vs all-classes {rows[0][1] ** 2:.3e} (ratio {s2 / rows[0][1] ** 2:.2f})")


class WeightSim(TorchFunctionMode):
    """fp16 rounding of the WEIGHT operand only, for the enabled classes (activations and everything else exact fp32)."""

    def __init__(self, wclass: dict, enabled):
        super().__init__()
        self.wclass, self.enabled = wclass, set(enabled)

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in (F.linear, F.conv2d) and self.wclass.get(id(args[1]), "cond") in self.enabled:
            a = list(args)
            a[1] = h(a[1])
            return func(*a, **kwargs)
        return func(*args, **kwargs)


def weights(which):
    """python tools/precision_sim.py weights [tiny|full32|full64]: which product classes carry the error of STORING the weights in fp16
    (the gap between the engine's 6.5-7.3e-4 against fp16-rounded oracle weights and 1.1-1.2e-3 against the reference's fp32 weights)."""
    torch.set_num_threads(8)
    if which == "tiny":
        import numpy as np
        cfg, hw = TINY, 16
        sd = {k: torch.from_numpy(np.asarray(v)).float() for k, v in recipe.state_dict(cfg, 0).items()}
    else:
        cfg, hw = UNetConfig(), int(which[4:])
        sd = random_state_dict(cfg, torch.device("cpu"), seed=3)
    wclass = {id(v): classify(k, v) for k, v in sd.items() if k.endswith(".weight") and v.dim() >= 2}
    wcls = [c for c in CLASSES if c not in ("qk", "pv")]
    inp = {k: torch.from_numpy(v) for k, v in recipe.synth_inputs(cfg, 1, hw, n_boxes=8, n_rel=3, seed=4321).items()}
    t = torch.full((1,), 481, dtype=torch.long)
    args = (sd, cfg, inp["x"].half().float(), t, inp["context"].half().float(), inp["relations"].half().float(), inp["boxes"], inp["masks"],
            inp["positive_embeddings"])
    with torch.no_grad():
        ref = unet_ref.unet_forward(*args)
        tol = 1e-4 + 1e-3 * ref.abs()
        rows = []
        for name, en in [("ALL weights fp16", wcls)] + [(c, (c,)) for c in wcls] + [("all but 1x1 convs", [c for c in wcls if c not in ("proj_in", "proj_out", "skip1x1")])]:
            t0 = time.time()
            with WeightSim(wclass, en):
                out = unet_ref.unet_forward(*args)
            d = out - ref
            r = float(d.norm() / ref.norm())
            rows.append((name, r))
            print(f"{name:18s} rel_l2={r:.3e}  rel_l2^2 share={r * r / (rows[0][1] ** 2) * 100:5.1f}%  outside rtol1e-3/atol1e-4: "
                  f"{float((d.abs() > tol).float().mean()) * 100:5.1f}%   ({time.time() - t0:.0f}s)", flush=True)


def strict(which):
    import torch.nn.functional as F_
    torch.set_num_threads(8)
    if which == "tiny":
        import numpy as np
        cfg, hw = TINY, 16
        sd = {k: torch.from_numpy(np.asarray(v)).float() for k, v in recipe.state_dict(cfg, 0).items()}
    else:
        cfg, hw = UNetConfig(), int(which[4:])
        sd = random_state_dict(cfg, torch.device("cpu"), seed=3)
    sd32 = sd
    sd = {k: (v.half().float() if v.dim() >= 2 else v) for k, v in sd.items()}
    wclass = {id(v): classify(k, v) for k, v in sd.items() if k.endswith(".weight") and v.dim() >= 2}
    wrela = {id(v): ("rela" if ".rela_fuse." in k else "other") for k, v in sd.items() if k.endswith(".weight") and v.dim() >= 2}
    inp = {k: torch.from_numpy(v) for k, v in recipe.synth_inputs(cfg, 1, hw, n_boxes=8, n_rel=3, seed=4321).items()}
    t = torch.full((1,), 481, dtype=torch.long)
    args = (sd, cfg, inp["x"].half().float(), t, inp["context"].half().float(), inp["relations"].half().float(), inp["boxes"], inp["masks"],
            inp["positive_embeddings"])
    with torch.no_grad():
        ref = unet_ref.unet_forward(*args)
        tol = 1e-4 + 1e-3 * ref.abs()
        print(f"# classes left on single-fp16 operands (all others split = exact here); reference = fp32 oracle on fp16-representable weights; |ref| rms {float(ref.pow(2).mean().sqrt()):.3f}")
        for en in (["qk", "pv"], ["qk"], ["pv"], ["qk", "pv", "qkv"], ["qkv", "cond"], ["qk", "pv", "qkv", "cond"], ["qk", "pv", "qkv", "attn_out"],
                   ["qk", "pv", "qkv", "attn_out", "ff_out"], ["qk", "pv", "qkv", "attn_out", "ff_out", "cond"],
                   ["qk", "pv", "qkv", "attn_out", "ff_out", "cond", "ff_in"]):
            with ClassSim(wclass, en):
                out = unet_ref.unet_forward(*args)
            d = out - ref
            print(f"unsplit: {'+'.join(en):48s} rel_l2={float(d.norm() / ref.norm()):.3e} outside rtol1e-3/atol1e-4: {float((d.abs() > tol).float().mean()) * 100:5.2f}%", flush=True)
        with ClassSim(wrela, ["rela"]):
            out = unet_ref.unet_forward(*args)
        d = out - ref
        print(f"unsplit: the Linear layers of rela_fuse only (1/30 weight)   rel_l2={float(d.norm() / ref.norm()):.3e} outside: {float((d.abs() > tol).float().mean()) * 100:5.2f}%", flush=True)
        # the floor: the fp32 oracle against itself in fp64 (unrounded weights and inputs)
        a32 = (sd32, cfg, inp["x"], t, inp["context"], inp["relations"], inp["boxes"], inp["masks"], inp["positive_embeddings"])
        r32 = unet_ref.unet_forward(*a32)
        te32, gn32 = unet_ref.timestep_embedding, unet_ref._group_norm
        unet_ref.timestep_embedding = lambda tt, dim, mp=10000.0: te32(tt, dim, mp).double()      # the same fp32 sinusoid values
        unet_ref._group_norm = lambda sd_, p, x, eps: F_.group_norm(x, 32, sd_[p + ".weight"], sd_[p + ".bias"], eps)
        try:
            r64 = unet_ref.unet_forward({k: v.double() for k, v in sd32.items()}, cfg, inp["x"].double(), t, inp["context"].double(),
                                        inp["relations"].double(), inp["boxes"], inp["masks"], inp["positive_embeddings"].double())
        finally:
            unet_ref.timestep_embedding, unet_ref._group_norm = te32, gn32
        d = r32.double() - r64
        print(f"fp32 oracle vs its fp64 evaluation: rel_l2={float(d.norm() / r64.norm()):.3e} max|err|={float(d.abs().max()):.2e} "
              f"outside: {float((d.abs() > 1e-4 + 1e-3 * r64.abs()).double().mean()) * 100:.3f}%")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "strict":
        return strict(sys.argv[2] if len(sys.argv) > 2 else "tiny")
    if len(sys.argv) > 1 and sys.argv[1] == "weights":
        return weights(sys.argv[2] if len(sys.argv) > 2 else "tiny")
    if len(sys.argv) > 1 and sys.argv[1] == "attribute":
        return attribute(sys.argv[2] if len(sys.argv) > 2 else "tiny")
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
