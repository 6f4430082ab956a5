#!/usr/bin/env python3
"""Groups a rocprofv3 kernel_stats.csv (tools/profile_round.sh: `bench.py --steps 1 --warmup 1` = 104 UNet forwards + 2 VAE decodes)
into kernel classes and prints ms per UNet forward.  usage: python tools/stats_breakdown.py profiles/r3_kernel_stats.csv [forwards]"""
import csv
import sys

path = sys.argv[1]
nf = float(sys.argv[2]) if len(sys.argv) > 2 else 104.0
import re


def g8(n, conv, s3):
    """gemm8_kernel<BM, BN, CONV, DBG[, S3]>: the conv flag is the THIRD template argument, the three-pass loop the fifth (round 6)"""
    m = re.search(r"gemm8_kernel<(\d+), (\d+), (true|false), (\d+)(?:, (true|false))?>", n)
    return bool(m) and (m.group(3) == "true") == conv and ((m.group(5) or "false") == "true") == s3


classes = [("8-wave conv, three-pass loop", lambda n: g8(n, True, True)),
           ("8-wave GEMM, three-pass loop", lambda n: g8(n, False, True)),
           ("8-wave conv (gemm8<.,true>)", lambda n: g8(n, True, False)),
           ("8-wave GEMM (gemm8<.,false>)", lambda n: "gemm8_kernel" in n),
           ("4-wave conv", lambda n: "gemm_kernel<" in n and ", true," in n),
           ("4-wave GEMM", lambda n: "gemm_kernel<" in n),
           ("skinny GEMM", lambda n: "gemm_skinny" in n),
           ("split-K reduce", lambda n: "splitk_reduce" in n),
           ("fused FeedForward", lambda n: "ff_fused" in n),
           ("split-fp16 attention (strict)", lambda n: "attn_split" in n),
           ("attention", lambda n: "attn_kernel" in n or "transpose_v" in n or "attention_small" in n),
           ("split_f32 (strict)", lambda n: "split_f32" in n),
           ("GroupNorm", lambda n: "gn_" in n),
           ("LayerNorm", lambda n: "layernorm" in n),
           ("relation cross-attention", lambda n: "rela_" in n),
           ("other HIP kernels of this repo", lambda n: "_GLOBAL__N_" in n or "anonymous namespace" in n),
           ("torch / runtime", lambda n: True)]
tot = {c: [0, 0.0] for c, _ in classes}
for r in csv.DictReader(open(path)):
    n = r["Name"]
    for c, f in classes:
        if f(n):
            tot[c][0] += int(r["Calls"])
            tot[c][1] += float(r["TotalDurationNs"])
            break
s = sum(v[1] for v in tot.values())
print(f"{'class':36s} {'launches/fwd':>12s} {'ms/fwd':>8s} {'share':>6s}")
for c, _ in classes:
    k, t = tot[c]
    print(f"{c:36s} {k / nf:12.1f} {t / nf * 1e-6:8.3f} {100 * t / s:5.1f}%")
print(f"{'total':36s} {sum(v[0] for v in tot.values()) / nf:12.1f} {s / nf * 1e-6:8.3f}   (includes the two VAE decodes of the run, ~0.3 ms per forward)")
