#!/usr/bin/env python
"""Build an A/B variant of the C-ABI library: the bf16 sources with extra -D flags on chosen files, linked to
tools/ablate/libwiw_<name>.so (git-ignored; travels to the GPU box).  Load it with WIW_LIB=tools/ablate/libwiw_<name>.so.

    python tools/build_variant.py trace ffn.hip:-DWIW_FFN_TRACE
    python tools/build_variant.py noge  ffn.hip:-DFFN_ABLATE=1
Files without a flag spec reuse the objects of the regular build (world-in-world_amd/build/)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wiw_amd  # noqa: E402,F401
from wiw_amd import build as B  # noqa: E402


def main():
    name = sys.argv[1]
    specs = {}
    for a in sys.argv[2:]:
        f, flags = a.split(":", 1)
        specs[f] = flags.split(",")
    B.build(verbose=False)
    out = os.path.join(ROOT, "tools", "ablate")
    os.makedirs(out, exist_ok=True)
    objs = []
    for src in B.SOURCES:
        if src in specs:
            obj = os.path.join(out, f"{name}_{src.replace('.hip', '.o')}")
            cmd = [B.HIPCC, *B.FLAGS, *specs[src], *B.EXTRA.get(src, []), "-c", os.path.join(B.CSRC, src), "-o", obj]
            subprocess.check_call(cmd)
        else:
            obj = os.path.join(B.HERE, "build", src.replace(".hip", ".o"))
        objs.append(obj)
    lib = os.path.join(out, f"libwiw_{name}.so")
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib])
    print(lib)


if __name__ == "__main__":
    main()
