"""Builds libchattts_amd.so (gfx950) in-tree with hipcc.  `python -m chattts_amd.build [--force]`.

An object is rebuilt when the sha256 of (its source, the shared headers, the flags) differs from the stamp written next to it --
content, not mtime: a checkout or a snapshot copy that rewrites timestamps neither hides an edit nor triggers a rebuild.  The
library records the host it was linked on (`csrc/.built_on`); `__graft_entry__.build()` forces a full rebuild when that is not this
host (or `CTTS_FORCE_BUILD=1`), so that "it builds" is shown from the sources there instead of inherited from shipped objects.
Translation units compile in parallel (hipcc is single-threaded per file)."""
from __future__ import annotations

import hashlib
import os
import socket
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libchattts_amd.so")
SOURCES = ["gemm.hip", "decode.hip", "decode32.hip", "decode32x.hip", "prefill.hip", "prefill32.hip", "prefill32x.hip", "gpt.hip", "codec.hip", "codec_gemm.hip", "dvae.hip", "capi.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result"]
HEADERS = [os.path.join(CSRC, "common.hpp"), os.path.join(CSRC, "kernels.hpp"), os.path.join(CSRC, "decode_dev.hpp"),
           os.path.join(HERE, "..", "include", "chattts_amd.h")]


_TOOLCHAIN = {}


def _toolchain(hipcc: str) -> bytes:
    """`hipcc --version`: part of every object's digest, so a toolchain upgrade rebuilds"""
    if hipcc not in _TOOLCHAIN:
        try:
            _TOOLCHAIN[hipcc] = subprocess.run([hipcc, "--version"], capture_output=True, timeout=60).stdout
        except (OSError, subprocess.SubprocessError):
            _TOOLCHAIN[hipcc] = b"unknown"
    return _TOOLCHAIN[hipcc]


def _digest(src: str, flags, hipcc: str = "") -> str:
    h = hashlib.sha256()
    h.update(_toolchain(hipcc) if hipcc else b"")
    for p in (src, *HEADERS):
        with open(p, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    h.update(" ".join(flags).encode())
    return h.hexdigest()


def _stamp(obj: str) -> str:
    try:
        with open(obj + ".sha", "r") as fh:
            return fh.read().strip()
    except OSError:
        return ""


def built_on() -> str:
    try:
        with open(os.path.join(CSRC, ".built_on"), "r") as fh:
            return fh.read().strip()
    except OSError:
        return ""


def build(force: bool = False, verbose: bool = True, variant: str = "", defines=()) -> str:
    """`variant` / `defines`: a second build of the same ABI with extra -D flags (probe builds for A/B runs through CTTS_LIB), objects
    under csrc/build_<variant>/, library csrc/libchattts_amd_<variant>.so."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    odir = os.path.join(CSRC, "build_" + variant) if variant else CSRC
    lib = os.path.join(CSRC, f"libchattts_amd_{variant}.so") if variant else LIB
    os.makedirs(odir, exist_ok=True)
    flags = [*FLAGS, *[f"-D{d}" for d in defines]]
    jobs, objs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(odir, s.replace(".hip", ".o"))
        objs.append(obj)
        dg = _digest(src, flags, hipcc)
        if force or not os.path.exists(obj) or _stamp(obj) != dg:
            jobs.append((src, obj, dg))

    def compile_one(job):
        src, obj, dg = job
        cmd = [hipcc, *flags, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(obj + ".sha", "w") as fh:
            fh.write(dg)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), max(1, (os.cpu_count() or 2) - 1), 10)) as pool:
            list(pool.map(compile_one, jobs))
    # the library carries the digests of the objects it was linked from (csrc/<lib>.sha): an interrupted or failed link after a
    # successful compile leaves a stale library that the next build must not take for current
    want = hashlib.sha256("\n".join(_stamp(o) for o in objs).encode()).hexdigest()
    if force or jobs or not os.path.exists(lib) or _stamp(lib) != want:
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(lib + ".sha", "w") as fh:
            fh.write(want)
        if not variant:
            with open(os.path.join(CSRC, ".built_on"), "w") as fh:
                fh.write(socket.gethostname())
    return lib


if __name__ == "__main__":
    # python -m chattts_amd.build [--force] [--variant NAME -DFOO=1 ...]
    var = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else ""
    print(build(force="--force" in sys.argv, variant=var, defines=[a[2:] for a in sys.argv if a.startswith("-D")]))
