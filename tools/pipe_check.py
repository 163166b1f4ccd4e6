#!/usr/bin/env python
"""Bit-level A/B of two builds of the library on GEMM / conv launches: prints one hash per (shape, repeat) — run it under two
WIW_LIB settings and diff the outputs.  The free-running K loop of gemm_huge.hip (PIPE) accumulates every output element in
the same K order as the eight-slot loop, so the two builds must agree bit for bit.

    python tools/pipe_check.py > a.txt;  WIW_LIB=tools/ablate/libwiw_nopipe.so python tools/pipe_check.py > b.txt;  diff a.txt b.txt
"""
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wiw_amd  # noqa: E402,F401
from wiw_amd import hip as H  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    hip = H.Hip(dev)
    g = torch.Generator(device="cpu").manual_seed(5)
    cases = [  # (M, N, K, mode, C1, H, W, T, tiled, res, geglu[, halo])
        (57344, 320, 768, H.A_CONV_T3, 256, 32, 64, 7, False, False, False),
        (57344, 320, 2304, H.A_CONV3X3, 256, 32, 64, 7, False, False, False),
        (64512, 640, 640, H.A_DENSE, 640, 0, 0, 0, True, True, False),
        (64512, 5120, 640, H.A_DENSE, 640, 0, 0, 0, True, False, True),
        (64512, 640, 2560, H.A_DENSE, 2560, 0, 0, 0, True, True, False),
        (258048, 320, 320, H.A_DENSE, 320, 0, 0, 0, True, True, False),
        (258048, 320, 960, H.A_CONV_T3, 320, 72, 128, 14, True, False, False),
        (16128, 1280, 3840, H.A_CONV_T3, 1280, 18, 32, 14, True, False, False),
        (129024, 640, 1280, H.A_DENSE, 1280, 0, 0, 0, True, False, False),     # 2 output tiles per block and more
        (258048, 320, 2880, H.A_CONV3X3, 320, 72, 128, 14, True, True, False, True),     # halo-staged 3x3, the served geometries
        (64512, 640, 5760, H.A_CONV3X3, 640, 36, 64, 14, True, False, False, True),
        (16128, 1280, 11520, H.A_CONV3X3, 1280, 18, 32, 14, True, True, False, True),    # tiles straddle frames
        (258048, 640, 5760, H.A_CONV3X3_UP, 640, 72, 128, 14, True, False, False, True),
        (516096, 320, 2880, H.A_CONV3X3, 320, 72, 128, 14, True, False, False, True),    # several output tiles per block
    ]
    for case in cases:
        (M, N, K, mode, C1, Hh, Ww, T, tiled, res, geglu), halo = case[:11], (len(case) > 11 and case[11])
        A = (torch.randn(M if mode != H.A_CONV3X3_UP else M // 4, C1, generator=g) * 0.5).to(torch.bfloat16).to(dev)
        W = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        Wt = H.TiledW(W) if tiled else W
        n_out = N // 2 if geglu else N
        out = torch.empty(M, n_out, dtype=torch.bfloat16, device=dev)
        kw = dict(M=M, N=N, K=K, C1=C1, mode=mode, bias=bias)
        if mode != H.A_DENSE:
            kw.update(H=Hh, Wd=Ww, T=T)
        if res:
            kw.update(res1=(torch.randn(M, N, generator=g)).to(torch.bfloat16).to(dev), ldr1=N, beta1=1.0)
        if geglu:
            kw.update(epilogue=H.EPI_GEGLU, n_out=n_out)
        if halo:
            kw.update(epilogue=H.K_HALO32)
        for rep in range(3):
            out.zero_()
            hip.gemm(A, Wt, out, **kw)
            torch.cuda.synchronize()
            o = out.cpu()
            fin = bool(torch.isfinite(o.float()).all())
            h = hashlib.sha1(o.view(torch.int16).numpy().tobytes()).hexdigest()[:16]
            print(f"M={M} N={N} K={K} mode={mode} tiled={int(tiled)} res={int(res)} geglu={int(geglu)} halo={int(bool(halo))} rep={rep}: {h} finite={fin}", flush=True)


if __name__ == "__main__":
    main()
