"""Golden-vector case table shared by tools/make_goldens.py (reference side, build container only)
and tests/test_oracle_golden.py (oracle side, runs anywhere).

Every case is data: a kind, integer hyper-parameters and seeds.  Inputs and weights are *recipe*
tensors (layoutllm_t2i_amd/recipe.py), i.e. pure functions of (name, shape, seed), so the fixtures
under tests/golden/ only hold the reference's OUTPUTS.
"""
from __future__ import annotations

import numpy as np

from layoutllm_t2i_amd import recipe
from layoutllm_t2i_amd.arch import TINY

CTX = 768
MO = 30


def rnd(tag, shape, seed=7):
    return recipe.normal(f"golden.{tag}", tuple(shape), seed)


def rela_boxes(variant: str, B: int):
    """Box sets that exercise attention.py:321-346: normal/overlap/border, degenerate `break`, null,
    clamp of x1>1, and the empty-slice (right<left) case that yields NaN in the reference."""
    boxes = np.zeros((B, MO, 4), np.float32)
    masks = np.zeros((B, MO), np.float32)
    if variant == "normal":
        base = [(0.0, 0.0, 0.5, 0.5), (0.25, 0.25, 1.0, 1.0), (0.6, 0.1, 0.9, 0.45), (0.13, 0.55, 0.41, 0.99)]
        for b in range(B):
            n = 4 - (b % 2)
            for i in range(n):
                boxes[b, i] = base[(i + b) % 4]
            masks[b, :n] = 1
    elif variant == "degenerate":
        # second box collapses to zero width at 8x8 -> break drops the third (valid-looking) box too
        for b in range(B):
            boxes[b, 0] = (0.1, 0.1, 0.6, 0.7)
            boxes[b, 1] = (0.50, 0.2, 0.55, 0.9) if b == 0 else (0.3, 0.3, 0.8, 0.8)
            boxes[b, 2] = (0.2, 0.5, 0.9, 0.9)
            masks[b, :3] = 1
    elif variant == "null":
        pass
    elif variant == "clamp":
        for b in range(B):
            boxes[b, 0] = (0.5, 0.4, 1.3, 1.2)       # x1,y1 > 1 -> clamped to w,h
            boxes[b, 1] = (0.0, 0.0, 1.0, 1.0)       # whole map
            masks[b, :2] = 1
    elif variant == "maskgap":
        # mask says 2 valid, a third box is present but masked out
        for b in range(B):
            boxes[b, 0] = (0.1, 0.2, 0.4, 0.6)
            boxes[b, 1] = (0.5, 0.5, 0.9, 0.8)
            boxes[b, 2] = (0.0, 0.0, 1.0, 1.0)
            masks[b, :2] = 1
    elif variant == "empty_slice":
        # xywh passed where ltrb is expected (interface.py:551-564 quirk): right < left
        for b in range(B):
            boxes[b, 0] = (0.1, 0.1, 0.6, 0.6)
            boxes[b, 1] = (0.7, 0.2, 0.2, 0.5) if b == 0 else (0.2, 0.2, 0.7, 0.5)
            masks[b, :2] = 1
    else:
        raise ValueError(variant)
    return boxes, masks


CASES = [
    dict(name="schedule_s10", kind="schedule", S=10),
    dict(name="schedule_s50", kind="schedule", S=50),
    dict(name="alpha_gen", kind="alpha_gen"),
    dict(name="temb_320", kind="timestep_embedding", dim=320, t=[1, 21, 500, 981]),
    dict(name="temb_64", kind="timestep_embedding", dim=64, t=[0, 7, 999]),
    dict(name="fourier", kind="fourier", B=2),
    dict(name="posnet", kind="position_net", B=2, n_boxes=3),
    dict(name="posnet_null", kind="position_net", B=2, n_boxes=0),
    dict(name="res_same", kind="res_block", cin=64, cout=64, te=256, hw=8, B=2),
    dict(name="res_widen", kind="res_block", cin=64, cout=128, te=256, hw=8, B=2),
    dict(name="res_concat", kind="res_block", cin=192, cout=64, te=256, hw=4, B=1),
    dict(name="self_attn_d40", kind="self_attn", C=320, heads=8, N=48, B=1),
    dict(name="self_attn_d16", kind="self_attn", C=64, heads=4, N=64, B=2),
    dict(name="cross_attn_d40", kind="cross_attn", C=320, heads=8, N=32, M=77, B=1),
    dict(name="cross_attn_d32", kind="cross_attn", C=128, heads=4, N=16, M=10, B=2),
    dict(name="ff_64", kind="ff", C=64, N=20, B=2),
    dict(name="gated_sa_s1", kind="gated_sa", C=64, heads=4, N=64, B=2, scale=1.0),
    dict(name="gated_sa_s0", kind="gated_sa", C=64, heads=4, N=64, B=2, scale=0.0),
    dict(name="gated_sa_s05", kind="gated_sa", C=64, heads=4, N=64, B=2, scale=0.5),
    dict(name="rela_normal", kind="rela", C=64, heads=4, hw=8, B=2, R=10, n_rel=3, boxes="normal"),
    dict(name="rela_degenerate", kind="rela", C=64, heads=4, hw=8, B=2, R=10, n_rel=2, boxes="degenerate"),
    dict(name="rela_null", kind="rela", C=64, heads=4, hw=8, B=2, R=10, n_rel=0, boxes="null"),
    dict(name="rela_clamp", kind="rela", C=64, heads=4, hw=8, B=2, R=10, n_rel=5, boxes="clamp"),
    dict(name="rela_maskgap", kind="rela", C=64, heads=4, hw=16, B=2, R=10, n_rel=4, boxes="maskgap"),
    dict(name="rela_empty_slice", kind="rela", C=64, heads=4, hw=8, B=2, R=10, n_rel=3, boxes="empty_slice"),
    dict(name="st_64", kind="spatial_transformer", C=64, heads=4, hw=8, B=2, scale=1.0, boxes="normal"),
    dict(name="st_64_s0", kind="spatial_transformer", C=64, heads=4, hw=8, B=2, scale=0.0, boxes="degenerate"),
    dict(name="down_64", kind="down", C=64, hw=8, B=2),
    dict(name="up_64", kind="up", C=64, hw=4, B=2),
    dict(name="unet_tiny_cond", kind="unet", B=2, hw=16, t=[981, 981], grounding="real", scale=1.0, sdconv=False),
    dict(name="unet_tiny_null", kind="unet", B=2, hw=16, t=[401, 401], grounding="null", scale=1.0, sdconv=False),
    dict(name="unet_tiny_s0_sd", kind="unet", B=2, hw=16, t=[21, 21], grounding="real", scale=0.0, sdconv=True),
    dict(name="plms_tiny", kind="plms", B=2, hw=16, S=10, guidance=7.5, alpha_type=[0.3, 0.0, 0.7]),
    dict(name="vae_tiny", kind="vae", B=2, hw=8),
]


def unet_inputs(case, cfg=TINY):
    d = recipe.synth_inputs(cfg, case["B"], case["hw"], n_boxes=4, n_rel=3, seed=4321)
    return d


def case_inputs(case):
    """Numpy inputs of a case (weights excluded)."""
    k = case["kind"]
    nm = case["name"]
    if k == "timestep_embedding":
        return dict(t=np.asarray(case["t"], np.int64))
    if k == "fourier":
        return dict(boxes=np.abs(recipe.uniform(f"golden.{nm}.boxes", (case["B"], MO, 4), 7)))
    if k == "position_net":
        B, nb = case["B"], case["n_boxes"]
        boxes = np.zeros((B, MO, 4), np.float32)
        masks = np.zeros((B, MO), np.float32)
        emb = np.zeros((B, MO, CTX), np.float32)
        if nb:
            u = np.abs(recipe.uniform(f"golden.{nm}.boxes", (B, nb, 4), 7))
            boxes[:, :nb] = np.sort(u.reshape(B, nb, 2, 2), axis=2).reshape(B, nb, 4)
            masks[:, :nb] = 1
            emb[:, :nb] = rnd(f"{nm}.emb", (B, nb, CTX))
        return dict(boxes=boxes, masks=masks, positive_embeddings=emb)
    if k == "res_block":
        B, hw = case["B"], case["hw"]
        return dict(x=rnd(f"{nm}.x", (B, case["cin"], hw, hw)), emb=rnd(f"{nm}.emb", (B, case["te"])))
    if k == "self_attn":
        return dict(x=rnd(f"{nm}.x", (case["B"], case["N"], case["C"])))
    if k == "cross_attn":
        return dict(x=rnd(f"{nm}.x", (case["B"], case["N"], case["C"])),
                    ctx=rnd(f"{nm}.ctx", (case["B"], case["M"], CTX)))
    if k == "ff":
        return dict(x=rnd(f"{nm}.x", (case["B"], case["N"], case["C"])))
    if k == "gated_sa":
        return dict(x=rnd(f"{nm}.x", (case["B"], case["N"], case["C"])),
                    objs=rnd(f"{nm}.objs", (case["B"], MO, CTX)))
    if k in ("rela", "spatial_transformer"):
        B, hw, C = case["B"], case["hw"], case["C"]
        boxes, masks = rela_boxes(case["boxes"], B)
        R, n_rel = case.get("R", 10), case.get("n_rel", 3)
        rel = np.zeros((B, R, CTX), np.float32)
        if n_rel:
            rel[:, :n_rel] = rnd(f"{nm}.rel", (B, n_rel, CTX))
        out = dict(relations=rel, boxes=boxes, masks=masks)
        if k == "rela":
            out["x"] = rnd(f"{nm}.x", (B, hw * hw, C))
        else:
            out["x"] = rnd(f"{nm}.x", (B, C, hw, hw))
            out["context"] = rnd(f"{nm}.context", (B, 77, CTX))
            out["objs"] = rnd(f"{nm}.objs", (B, MO, CTX))
        return out
    if k in ("down", "up"):
        return dict(x=rnd(f"{nm}.x", (case["B"], case["C"], case["hw"], case["hw"])))
    if k in ("unet", "plms"):
        return unet_inputs(case)
    if k == "vae":
        return dict(z=rnd(f"{nm}.z", (case["B"], 4, case["hw"], case["hw"])) * np.float32(0.18215 * 2.0))
    return {}
