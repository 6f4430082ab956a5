"""Diagnostic for one attention case (GPU box): python tools/attn_case_probe.py d H Nq Nk B
Runs tests/test_gpu_kernels.py::test_attention_prescaled_q's data through both running-max paths (padding column / FMA), lists the
elements outside rtol 2e-3 / atol 2e-4*scale, and re-derives the worst one on the CPU with the kernel's ALGORITHM (64-key tiles, fp32 online
softmax, fp16-representable running max, deferred rescale at 2^8, P rounded to fp16, fp32 accumulation): if that matches the kernel, the
deviation is the fp16 rounding of P, not an implementation defect."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_kernels as T
from layoutllm_t2i_amd import ops
from layoutllm_t2i_amd._lib import init_device
from layoutllm_t2i_amd.weights import q_fold
init_device(0)
d, H, Nq, Nk, B = (int(a) for a in sys.argv[1:6]) if len(sys.argv) >= 6 else (8, 2, 154, 1052, 2)
C = H * d; c = q_fold(d)
q = T.rnd(f"pq{d}{Nq}", (B, Nq, C)); q[:, 0] *= 30.0
k = T.rnd(f"pk{d}{Nk}", (B, Nk, C)); k[:, Nk - 3] = q[:, 1] * 4.0
qs, qd = T.h16(q * c); k, kd = T.h16(k); v, vd = T.h16(T.rnd(f"pv{d}{Nk}", (B, Nk, C)))
vt = torch.full((B, H, d, ops.vt_ld(Nk)), float("nan"), dtype=torch.float16, device=T.DEV)
ops.transpose_v(vd, Nk * C, C, vt, B, H, d, Nk)
ref = T._attn_ref(qs / c, k, v, H).float()
for opt29 in (1, 0):
    ops.set_option(29, opt29)
    out = torch.empty(B, Nq, C, dtype=torch.float16, device=T.DEV)
    ops.attention(qd, Nq * C, C, kd, Nk * C, C, vt, out, Nq * C, C, B, H, d, Nq, Nk, 123.0, q_prescaled=True)
    o = out.float().cpu()
    err = (o - ref).abs(); scale = max(1.0, float(ref.abs().max())); tol = 2e-4 * scale + 2e-3 * ref.abs()
    bad = (err > tol).nonzero()
    print(f"opt29={opt29}: max err {err.max():.3e}, violations {len(bad)}")
    for b_, q_, c_ in bad.tolist()[:5]:
        print(f"   (b={b_}, q={q_}, c={c_}) ref={ref[b_, q_, c_]:+.6f} out={o[b_, q_, c_]:+.6f} err={err[b_, q_, c_]:.3e} tol={tol[b_, q_, c_]:.3e}")
ops.set_option(29, 1)
# emulation: exact fp32 logits, P = fp16(2^(s - m)), l = sum of the ROUNDED P, out = fp16(sum P V / l)
qh = qs.view(B, Nq, H, d).permute(0, 2, 1, 3).double(); kh = k.view(B, Nk, H, d).permute(0, 2, 1, 3).double(); vh = v.view(B, Nk, H, d).permute(0, 2, 1, 3).double()
s = qh @ kh.transpose(-1, -2)                      # exp2 units (prescaled)
m = s.max(-1, keepdim=True).values
for shift in (0.0, 4.0, 8.0):
    p = torch.exp2(s - m + shift).to(torch.float16).double()
    e = ((p @ vh) / p.sum(-1, keepdim=True)).permute(0, 2, 1, 3).reshape(B, Nq, C).to(torch.float16).float()
    err = (e - ref).abs()
    print(f"emulated fp16 P (reference value m - {shift}): max err {err.max():.3e}, violations {(err > tol).sum().item()}")

import numpy as np
def algo(b, qi, h, padmax, defer=8.0):
    f32 = np.float32
    qv = qs[b, qi, h*d:(h+1)*d].double().numpy(); K = k[b, :, h*d:(h+1)*d].double().numpy(); V = v[b, :, h*d:(h+1)*d].double().numpy()
    m = f32(0.0); o = np.zeros(d, f32); l = f32(0.0)
    for t in range((Nk + 63) // 64):
        ks = K[t*64:(t+1)*64]; vs = V[t*64:(t+1)*64]
        s_ = ((ks @ qv) - float(m)).astype(f32)
        tmax = s_.max()
        if t == 0 or tmax > defer:
            inc = tmax if t == 0 else max(tmax, f32(0))
            if padmax: inc = f32(np.float16(m + inc)) - m
            alpha = f32(1.0) if t == 0 else f32(2.0 ** float(-inc))
            m = f32(m + inc); o = (o * alpha).astype(f32); l = f32(l * alpha); s_ = (s_ - inc).astype(f32)
        p = np.exp2(s_.astype(np.float64)).astype(np.float16).astype(np.float64)
        o = (o + (p @ vs)).astype(f32); l = f32(l + p.sum())
    return (o / l).astype(np.float16).astype(np.float32)
ops.set_option(29, 1)
out = torch.empty(B, Nq, C, dtype=torch.float16, device=T.DEV)
ops.attention(qd, Nq * C, C, kd, Nk * C, C, vt, out, Nq * C, C, B, H, d, Nq, Nk, 123.0, q_prescaled=True)
o = out.float().cpu(); err = (o - ref).abs()
b_, q_, c_ = [int(x) for x in np.unravel_index(int(((err - tol).argmax())), err.shape)]
h_ = c_ // d
em = algo(b_, q_, h_, True)
print(f"worst element relative to its tolerance: (b={b_}, q={q_}, c={c_}) kernel {o[b_, q_, c_]:+.7f}  CPU emulation of the algorithm {em[c_ - h_ * d]:+.7f}  reference {ref[b_, q_, c_]:+.7f}  tol {tol[b_, q_, c_]:.3e}")
print("kernel == emulation on that (query, head):", bool(np.array_equal(o[b_, q_, h_*d:(h_+1)*d].numpy(), em)))
