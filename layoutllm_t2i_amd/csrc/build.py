"""Builds libgligen_hip.so (gfx950) from the .hip sources in this directory, in-tree.

    python -m layoutllm_t2i_amd.csrc.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  The resulting .so is git-ignored but travels to the
GPU box with the repo snapshot.  No torch headers: the library only depends on the HIP runtime.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
REPO = os.path.dirname(PKG)
SOURCES = ["gemm_conv.hip", "gemm8.hip", "attention.hip", "norms.hip", "rela.hip", "misc.hip", "engine.hip", "reward.hip", "ff_fused.hip", "clip.hip", "preprocess.hip", "vae_engine.hip"]
LIB = os.path.join(PKG, "libgligen_hip.so")
OBJDIR = os.path.join(HERE, "build")


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm's hipcc to build the gfx950 kernels)")


def _newer(a: str, b: str) -> bool:
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force: bool = False, verbose: bool = True, variant: str = "", defines=()) -> str:
    """``variant`` / ``defines``: a same-box A/B build of compile-time alternatives -- objects under build_<variant>/, library
    libgligen_hip_<variant>.so next to the product library (selected at run time with GLIGEN_HIP_LIB, see _lib.py); the product
    build takes neither."""
    hipcc = _hipcc()
    OBJDIR = os.path.join(HERE, "build_" + variant) if variant else globals()["OBJDIR"]
    LIB = os.path.join(PKG, f"libgligen_hip_{variant}.so") if variant else globals()["LIB"]
    os.makedirs(OBJDIR, exist_ok=True)
    deps = [os.path.join(HERE, "common.h"), os.path.join(HERE, "gemm_shared.h"), os.path.join(HERE, "opts.h"), os.path.join(REPO, "include", "gligen_hip.h")]
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(REPO, "include"), "-I" + HERE, *["-D" + d for d in defines]]

    # per-file extras: attention keeps MFMA results in VGPRs (no v_accvgpr_read/write round trips in the
    # softmax, which is VALU-bound; measured 223 -> 0 AGPR moves per 64-key tile)
    extra = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}

    def compile_one(src: str) -> str:
        s = os.path.join(HERE, src)
        o = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        if force or _newer(s, o) or any(_newer(d, o) for d in deps) or _newer(os.path.abspath(__file__), o):
            cmd = [hipcc, *flags, *extra.get(src, []), "-c", s, "-o", o]
            if verbose:
                print("[build]", " ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return o

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    if force or any(_newer(o, LIB) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    # python -m layoutllm_t2i_amd.csrc.build [--force] [--variant NAME -DX=Y ...]
    _v = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else ""
    print(build(force="--force" in sys.argv, variant=_v, defines=[a[2:] for a in sys.argv if a.startswith("-D")]))
