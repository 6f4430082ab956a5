"""PLMS sampler with the reference's call surface (GLIGEN/ldm/models/diffusion/plms.py:10-163).

    PLMSSampler(diffusion, model, alpha_generator_func=..., set_alpha_scale=...)
        .sample(S, shape, input, uc, guidance_scale, mask=None, x0=None) -> latent

Same loop, same side effects (per-step fuser scale via set_alpha_scale, permanent SD first-conv
switch on every scale-0 step, step-0 double evaluation, Adams-Bashforth history of <= 3 eps), but:

  * cond + uncond run as ONE 2B-sized UNet evaluation (result-identical per sample): the
    conditioning batch is [real grounding + prompt ; null grounding + ""], hoisted once per image;
  * the CFG combine and the x_prev update are two tiny fused HIP kernels on fp32 latents, evaluated
    in the reference's operation order (bit-identical to torch fp32 given the same eps);
  * the schedule tables are computed once per S.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch

from . import host, ops


class PLMSSampler(object):
    def __init__(self, diffusion, model, schedule="linear", alpha_generator_func=None, set_alpha_scale=None):
        self.diffusion = diffusion
        self.model = model
        self.device = model.device if hasattr(model, "device") else diffusion.betas.device
        self.ddpm_num_timesteps = diffusion.num_timesteps
        self.schedule = schedule
        self.alpha_generator_func = alpha_generator_func
        self.set_alpha_scale = set_alpha_scale
        self.mirror_rng = True     # plms.py:138 draws randn_like(x) * 0 on every update
        self._sched_S = None

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=False):
        if ddim_eta != 0:
            raise ValueError('ddim_eta must be 0 for PLMS')
        if ddim_discretize != "uniform":
            raise NotImplementedError(ddim_discretize)
        if self._sched_S == ddim_num_steps:
            return
        acp = self.diffusion.alphas_cumprod.detach().cpu().numpy().astype(np.float32)
        assert acp.shape[0] == self.ddpm_num_timesteps, 'alphas have to be defined for each timestep'
        s = host.make_schedule(ddim_num_steps, acp)
        self.ddim_timesteps = s["ddim_timesteps"]
        self.ddim_alphas = s["ddim_alphas"]
        self.ddim_alphas_prev = s["ddim_alphas_prev"]
        self.ddim_sqrt_one_minus_alphas = s["ddim_sqrt_one_minus_alphas"]
        self.ddim_sigmas = s["ddim_sigmas"]
        self._sched = s
        self._sched_S = ddim_num_steps

    @torch.no_grad()
    def sample(self, S, shape, input, uc=None, guidance_scale=1, mask=None, x0=None):
        self.make_schedule(ddim_num_steps=S)
        return self.plms_sampling(shape, input, uc, guidance_scale, mask=mask, x0=x0)

    @torch.no_grad()
    def plms_sampling(self, shape, input, uc=None, guidance_scale=1, mask=None, x0=None):
        if mask is not None:
            raise NotImplementedError("inpainting masks are not on the layout-to-image path (interface.py:512)")
        model, eng = self.model, self.model.engine
        dev = self.device
        b = shape[0]
        img = input["x"]
        if img is None:
            img = torch.randn(shape, device=dev)
            input["x"] = img
        x = img.to(dev, torch.float32).contiguous().clone()
        side = x.shape[-1]
        n = x.numel()

        time_range = np.flip(self.ddim_timesteps)
        total_steps = self.ddim_timesteps.shape[0]
        alphas = self.alpha_generator_func(len(time_range)) if self.alpha_generator_func is not None else None

        # ---- conditioning, once per image: [cond ; uncond] when CFG is active (plms.py:115-124)
        cfg_on = uc is not None and guidance_scale != 1
        g = model.grounding_of(input)
        f32 = lambda t: t.to(dev, torch.float32)
        if cfg_on:
            if model.grounding_tokenizer_input is None:
                raise RuntimeError("model.grounding_tokenizer_input is not set (interface.py:370)")
            gn = model.grounding_tokenizer_input.get_null_input()
            ctx = torch.cat([f32(input["context"]), f32(uc)], 0)
            rel = torch.cat([f32(input["relations"])] * 2, 0)
            grounding = {k: torch.cat([f32(g[k]), f32(gn[k])], 0) for k in ("boxes", "masks", "positive_embeddings")}
            reps = 2
        else:
            ctx, rel = f32(input["context"]), f32(input["relations"])
            grounding = {k: f32(g[k]) for k in ("boxes", "masks", "positive_embeddings")}
            reps = 1
        model.set_conditioning(ctx, rel, grounding, side, key=None)

        ring = [eng.buf(f"plms.e{j}", tuple(x.shape), torch.float32) for j in range(4)]
        x_a = eng.buf("plms.x", tuple(x.shape), torch.float32)    # running latent (the engine copies it per forward)
        x_mid = eng.buf("plms.xmid", tuple(x.shape), torch.float32)
        x_a.copy_(x)
        old: List[torch.Tensor] = []      # newest last, <= 3 entries; tensors are slots of `ring`
        free = list(ring)

        def step_eval(x_eval, t_val, e_out, e_terms, coefs, div, index, x_out):
            """one gl_plms_step: UNet (2B batch) -> CFG combine into e_out -> e' -> x_out = update of x_a"""
            sq_at, s1m, sq_ap, dirc = host.step_coefs(self._sched, index)
            if self.mirror_rng:
                torch.randn_like(x_a)          # plms.py:138 draws noise * sigma (= 0) on every update: keep the RNG stream aligned
            eng.plms_step(x_eval, x_a, x_out, e_out, e_terms, coefs, div, float(t_val), reps, float(guidance_scale),
                          model.fuser_scale, model.use_sd_conv, sq_at, s1m, sq_ap, dirc)

        for i, step in enumerate(time_range):
            if alphas is not None:
                self.set_alpha_scale(model, alphas[i])
                if alphas[i] == 0:
                    model.restore_first_conv_from_SD()
            index = total_steps - i - 1
            t_next = time_range[min(i + 1, len(time_range) - 1)]
            e_t = free.pop()
            if len(old) == 0:
                # step 0 evaluates twice (plms.py:144-150): x_mid from e_t alone, then e' = (e_t + e(x_mid, t_next)) / 2
                step_eval(x_a, int(step), e_t, [e_t], (1.0,), 1.0, index, x_mid)
                e_next = free[-1]                                   # scratch slot, not kept
                coefs, div = host.PLMS_COEFS[0]
                step_eval(x_mid, int(t_next), e_next, [e_t, e_next], coefs, div, index, x_a)
            else:
                coefs, div = host.PLMS_COEFS[min(len(old), 3)]
                step_eval(x_a, int(step), e_t, [e_t] + old[::-1][:len(coefs) - 1], coefs, div, index, x_a)
            old.append(e_t)
            if len(old) >= 4:
                free.append(old.pop(0))
        input["x"] = x_a.clone()
        return input["x"]
