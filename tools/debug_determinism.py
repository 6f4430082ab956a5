import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import golden_cases as gc
from layoutllm_t2i_amd import recipe, host, ops
from layoutllm_t2i_amd.arch import TINY
from layoutllm_t2i_amd.interface import denoise
from layoutllm_t2i_amd.model import GroundingNetInput, LatentDiffusion, UNetModel
from oracle import plms_ref
DEV = "cuda:0"; T = torch.from_numpy
case = next(c for c in gc.CASES if c["name"] == "plms_tiny")
sd = recipe.state_dict(TINY, 0)
model = UNetModel(TINY, sd, device=DEV, sd_first_conv=recipe.sd_first_conv(TINY, 0))
model.grounding_tokenizer_input = GroundingNetInput()
inp = {a: T(v) for a, v in gc.case_inputs(case).items()}
diffusion = LatentDiffusion(device=DEV)
batch = dict(boxes=inp["boxes"], masks=inp["masks"], text_embeddings=inp["positive_embeddings"])
eng = model.engine
log = []
orig = eng.forward
def rec(x_lat, t, fuser_scale=1.0, sd_conv=False, reps=1, eps_out=None):
    out = orig(x_lat, t, fuser_scale, sd_conv, reps, eps_out)
    log.append((x_lat.detach().cpu().clone(), float(t), float(fuser_scale), bool(sd_conv), out.detach().cpu().clone()))
    return out
eng.forward = rec
def run():
    model.first_conv_type = "GLIGEN"
    return denoise((model, None, None, diffusion, {}), inp["context"], inp["uc"], inp["relations"], batch, inp["x"].to(DEV), case["alpha_type"], case["guidance"], steps=case["S"]).cpu()
a = run(); la = list(log); log.clear()
b = run(); lb = list(log); log.clear()
print("denoise twice bitwise equal:", torch.equal(a, b))
for i, (p, q) in enumerate(zip(la, lb)):
    if not (torch.equal(p[0], q[0]) and torch.equal(p[4], q[4])):
        print(" first diff at forward", i, "x equal", torch.equal(p[0], q[0]), "eps equal", torch.equal(p[4], q[4])); break
z = torch.zeros_like; cat = lambda u, v: torch.cat([u, v], 0)
eng.set_conditioning(cat(inp["context"], inp["uc"]), cat(inp["relations"], inp["relations"]), cat(inp["boxes"], z(inp["boxes"])),
                     cat(inp["masks"], z(inp["masks"])), cat(inp["positive_embeddings"], z(inp["positive_embeddings"])), 16)
state = dict(sd=False)
def eps_fn(x, t, i, alpha):
    if alpha == 0: state["sd"] = True
    e2 = eng.forward(x.to(DEV), float(t[0]), float(alpha), state["sd"], 2).cpu()
    return e2[2:] + case["guidance"] * (e2[:2] - e2[2:])
ref = plms_ref.plms_sample(eps_fn, inp["x"], case["S"], case["alpha_type"])
lc = list(log)
print("sampler vs oracle-loop equal:", torch.equal(a, ref), float((a - ref).abs().max()))
for i, (p, q) in enumerate(zip(la, lc)):
    ex, ee = torch.equal(p[0], q[0]), torch.equal(p[4], q[4])
    print(f" fwd {i}: t {p[1]} {q[1]} scale {p[2]} {q[2]} sd {p[3]} {q[3]} x_equal {ex} (max {float((p[0]-q[0]).abs().max()):.3e}) eps_equal {ee}")
    if not (ex and ee): break

print("---- step-1 update dissection")
g = case["guidance"]
x2 = la[2][0]
def cfg_cpu(e2): return e2[2:] + g * (e2[:2] - e2[2:])
e_t, e_old = cfg_cpu(la[2][4]), cfg_cpu(la[0][4])
ed = torch.empty_like(e_t, device=DEV); ops.cfg_combine(la[2][4].to(DEV), g, ed)
print("cfg equal", torch.equal(ed.cpu(), e_t))
sched = host.make_schedule(10, host.alphas_cumprod())
index = 8
sq_at, s1m, sq_ap, dirc = host.step_coefs(sched, index)
xp = torch.empty_like(x2, device=DEV)
ops.plms_update(x2.to(DEV), e_t.to(DEV), [e_old.to(DEV)], (3.0, -1.0), 2.0, sq_at, s1m, sq_ap, dirc, xp)
b = 2
a_t = torch.full((b,1,1,1), float(sched["ddim_alphas"][index])); a_prev = torch.full((b,1,1,1), float(sched["ddim_alphas_prev"][index]))
s1 = torch.full((b,1,1,1), float(sched["ddim_sqrt_one_minus_alphas"][index]))
ep = (3 * e_t - e_old) / 2
pred = (x2 - s1 * ep) / a_t.sqrt()
ref = a_prev.sqrt() * pred + (1.0 - a_prev).sqrt() * ep
print("update equal", torch.equal(xp.cpu(), ref), float((xp.cpu()-ref).abs().max()))
print("x at fwd3 (sampler) == kernel", torch.equal(la[3][0], xp.cpu()), " == cpu ref", torch.equal(la[3][0], ref), " oracle-flow == cpu ref", torch.equal(lc[3][0], ref))
print("coefs", sq_at, float(a_t.sqrt()[0,0,0,0]), s1m, float(s1[0,0,0,0]), sq_ap, float(a_prev.sqrt()[0,0,0,0]), dirc, float((1.0-a_prev).sqrt()[0,0,0,0]))
