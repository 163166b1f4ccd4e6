"""Build libwiwsvd.so and libwiwsvd_f16.so (hand-written gfx950 HIP kernels + C ABI) in-tree with hipcc.

The same sources are compiled twice: bf16 storage (libwiwsvd.so, the default) and IEEE fp16 storage (-DWIW_F16,
libwiwsvd_f16.so: the reference's served default dtype); `wiw_dtype()` tells them apart.  hipcc cross-compiles for
gfx950 without a GPU; the .so files are git-ignored but travel to the GPU box with the gpurun snapshot.
Usage: python world-in-world_amd/build.py [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libwiwsvd.so")
LIB_F16 = os.path.join(HERE, "libwiwsvd_f16.so")
VARIANTS = [(LIB, "build", []), (LIB_F16, "build_f16", ["-DWIW_F16=1"])]
SOURCES = ["gemm.hip", "gemm_huge.hip", "ffn.hip", "ffn32.hip", "attention.hip", "attention32.hip", "cross_attn.hip", "temporal.hip", "clip.hip", "norm.hip", "elementwise.hip", "vae.hip", "train.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# per-file extras: the attention softmax never sees NaNs (infinities are used and preserved); dropping NaN
# canonicalisation removes one v_max per fmaxf in its inner loop
EXTRA = {"attention.hip": ["-fno-honor-nans"], "attention32.hip": ["-fno-honor-nans", "-fno-slp-vectorize"],
         "ffn32.hip": ["-fno-slp-vectorize"]}


def _stale(lib: str) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "wiw_svd.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every stale variant (all object files of all variants in parallel); returns the bf16 library path."""
    todo = [v for v in VARIANTS if force or _stale(v[0])]
    procs = []
    for lib, bdir, defs in todo:
        os.makedirs(os.path.join(HERE, bdir), exist_ok=True)
        for src in SOURCES:
            obj = os.path.join(HERE, bdir, src.replace(".hip", ".o"))
            cmd = [HIPCC, *FLAGS, *defs, *EXTRA.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    for lib, bdir, _ in todo:
        objs = [os.path.join(HERE, bdir, src.replace(".hip", ".o")) for src in SOURCES]
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
