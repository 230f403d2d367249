"""Builds libhudiff_hip.so (gfx950) in-tree with hipcc.  `python -m hudiff_amd.build`."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libhudiff_hip.so")
SOURCES = [os.path.join(CSRC, "hd_api.hip")]
DEPS = SOURCES + [os.path.join(CSRC, "hd_kernels.hip.h"), os.path.join(CSRC, "hd_tail_fused.hip.h"),
                  os.path.join(CSRC, "hd_attn_fused.hip.h"), os.path.join(CSRC, "hd_chain.hip.h"),
                  os.path.join(os.path.dirname(HERE), "include", "hudiff_hip.h")]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in DEPS)


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every HIP source for gfx950 into hudiff_amd/libhudiff_hip.so (cross-compiles without a GPU)."""
    if not force and not needs_build():
        return LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wno-unused-result", "-Wno-unused-value",
           # (hd_chain.hip.h: its fully unrolled LayerNorm / split passes exceed LLVM's default 16 K-instruction bound for `#pragma unroll`)
           "-mllvm", "-pragma-unroll-threshold=200000", *os.environ.get("HUDIFF_CXXFLAGS", "").split(), *SOURCES, "-o", LIB]
    if verbose:
        print("[hudiff_amd.build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
