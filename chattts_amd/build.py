"""Builds libchattts_amd.so (gfx950) in-tree with hipcc.  `python -m chattts_amd.build`."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libchattts_amd.so")
SOURCES = ["gemm.hip", "decode.hip", "decode32.hip", "prefill.hip", "prefill32.hip", "gpt.hip", "codec.hip", "codec_gemm.hip", "dvae.hip", "capi.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result"]


def _newer(a: str, b: str) -> bool:
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force: bool = False, verbose: bool = True, variant: str = "", defines=()) -> str:
    """`variant` / `defines`: a second build of the same ABI with extra -D flags (probe builds for A/B runs through CTTS_LIB), objects
    under csrc/build_<variant>/, library csrc/libchattts_amd_<variant>.so."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, h) for h in ("common.hpp", "kernels.hpp")] + [os.path.join(HERE, "..", "include", "chattts_amd.h")]
    odir = os.path.join(CSRC, "build_" + variant) if variant else CSRC
    lib = os.path.join(CSRC, f"libchattts_amd_{variant}.so") if variant else LIB
    os.makedirs(odir, exist_ok=True)
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(odir, s.replace(".hip", ".o"))
        if force or _newer(src, obj) or any(_newer(h, obj) for h in headers):
            cmd = [hipcc, *FLAGS, *[f"-D{d}" for d in defines], "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(obj)
    if force or any(_newer(o, lib) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    # python -m chattts_amd.build [--force] [--variant NAME -DFOO=1 ...]
    var = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else ""
    print(build(force="--force" in sys.argv, variant=var, defines=[a[2:] for a in sys.argv if a.startswith("-D")]))
