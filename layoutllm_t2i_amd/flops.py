"""Algorithmic FLOP count of one UNet forward (2 x multiply-accumulates of every conv / Linear / attention product),
derived from the block plan -- the figure ``bench.py``'s roofline uses.  Matches SURVEY 8d at the reference shapes:
64x64 latent: 1.1477 TFLOP per sample with the fuser, 0.8141 with it skipped; 96x96: 3.2495 TFLOP.

Counted as the REFERENCE executes it (openaimodel.py:413-459): PositionNet, fuser.linear and the cross-attention K/V
projections are per-forward work there (this build hoists them; they are <0.2 % of the total)."""
from __future__ import annotations

from .arch import UNetConfig, build_plan


def unet_forward_flops(cfg: UNetConfig, hw: int, fuser_on: bool = True, n_ctx: int = 77, n_rel: int = 10) -> float:
    plan = build_plan(cfg)
    mo, ctx, te = cfg.max_objs, cfg.context_dim, cfg.time_embed_dim
    f = 0.0
    lin = lambda m, k, n: 2.0 * m * k * n

    def attn(nq, nk, C):      # QK^T + PV over all heads
        return 2.0 * 2.0 * nq * nk * C

    f += lin(1, cfg.model_channels, te) + lin(1, te, te)
    f += lin(mo, cfg.pos_in_dim + cfg.position_dim, 512) + lin(mo, 512, 512) + lin(mo, 512, cfg.pos_out_dim)
    side = hw

    def layer(l, side):
        n = side * side
        g = 0.0
        if l.kind == "conv_in":
            g += lin(n, 9 * l.cin, l.cout)
        elif l.kind == "res":
            g += lin(n, 9 * l.cin, l.cout) + lin(n, 9 * l.cout, l.cout) + lin(1, te, l.cout)
            if l.cin != l.cout:
                g += lin(n, l.cin, l.cout)
        elif l.kind == "down":
            g += lin(n // 4, 9 * l.cin, l.cout)
        elif l.kind == "up":
            g += lin(n * 4, 9 * l.cin, l.cout)
        elif l.kind == "st":
            C = l.cin
            g += 2 * lin(n, C, C)                                        # proj_in / proj_out
            g += 3 * lin(n, C, C) + attn(n, n, C) + lin(n, C, C)         # attn1
            if fuser_on:
                g += lin(mo, ctx, C)                                     # fuser.linear
                g += 3 * lin(n + mo, C, C) + attn(n + mo, n + mo, C) + lin(n + mo, C, C)
                g += lin(n, C, 8 * C) + lin(n, 4 * C, C)                 # fuser.ff
            g += lin(mo, C, C) + 2 * lin(n_rel, ctx, C) + attn(mo, n_rel, C) + lin(mo, C, C)   # rela_fuse.attn
            g += lin(mo, C, 8 * C) + lin(mo, 4 * C, C)                   # rela_fuse.ff
            g += lin(n, C, C) + 2 * lin(n_ctx, ctx, C) + attn(n, n_ctx, C) + lin(n, C, C)      # attn2
            g += lin(n, C, 8 * C) + lin(n, 4 * C, C)                     # ff
        return g

    for b in plan.input_blocks:
        for l in b.layers:
            f += layer(l, side)
            if l.kind == "down":
                side //= 2
    for l in plan.middle.layers:
        f += layer(l, side)
    for b in plan.output_blocks:
        for l in b.layers:
            f += layer(l, side)
            if l.kind == "up":
                side *= 2
    f += lin(side * side, 9 * plan.out_channels_last, cfg.out_channels)
    return f
