#!/usr/bin/env python
"""Time individual GEMM / conv launches of libwiwsvd.so (HIP events, random data) — used to iterate on
the kernel and to attach rocprofv3 PMC counters to ONE shape.

    python tools/gemm_probe.py M,N,K[,mode[,epi]] ...      e.g.  258048,2560,320,0,1   (GEGLU)
env: ITERS, TILED=1 (hip.TiledW weights), RES=1 (residual operand), SPLITK=n (WiwGemmArgs.splitk), KCMAJOR=1 (WIW_K_CMAJOR on conv modes)
mode: 0 dense 1 conv3x3 2 s2 3 up 4 temporal (conv shapes use H=72,W=128-like factorisation of M)
"""
import math
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
    iters = int(os.environ.get("ITERS", "20"))
    for spec in sys.argv[1:]:
        parts = [int(v) for v in spec.split(",")]
        M, N, K = parts[:3]
        mode = parts[3] if len(parts) > 3 else 0
        epi = parts[4] if len(parts) > 4 else 0
        taps = {0: 1, 1: 9, 2: 9, 3: 9, 4: 3}[mode]
        C1 = K // taps
        C2 = int(os.environ.get("SC", "0")) if mode == 1 else 0   # fused shortcut segment behind the taps: K = 9 * C1 + C2
        if C2:
            C1 = (K - C2) // 9
        rows_in = M * 4 if mode == 2 else (M // 4 if mode == 3 else M)
        A = torch.randn(rows_in, C1, device=dev).to(torch.bfloat16)
        W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
        if os.environ.get("TILED"):
            W = H.TiledW(W)
        bias = torch.randn(N, device=dev)
        if mode and os.environ.get("KCMAJOR"):      # channel-block-major K walk (random weights: no re-ordering needed for timing)
            epi |= H.K_CMAJOR
        if mode in (1, 3) and os.environ.get("HALO"):    # halo-staged 3x3 kernel (needs TILED=1 and the served geometry below)
            epi |= H.K_HALO32
        kw = dict(M=M, N=N, K=K, C1=C1, mode=mode, bias=bias, epilogue=epi, splitk=int(os.environ.get("SPLITK", "1")))
        if C2:
            kw.update(A2=torch.randn(M, C2, device=dev).to(torch.bfloat16), C2=C2)
        if mode:
            frames = 28 if M % 28 == 0 else 1
            hw = M // frames
            h = int(math.sqrt(hw / 2)) if mode != 4 else 1
            while hw % h:
                h -= 1
            if os.environ.get("SERVED") and mode != 4:     # the served latent geometry: 72 x 128 / 36 x 64 / 18 x 32 / 9 x 16
                h = {9216: 72, 2304: 36, 576: 18, 144: 9}.get(hw, h)
            kw.update(H=h, Wd=hw // h, T=14 if frames % 14 == 0 else 1)
        if epi & H.EPI_GEGLU:
            out = torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev)
            kw["n_out"] = N // 2
        else:
            out = torch.empty(M, N, dtype=torch.float32 if epi & H.EPI_OUT_F32 else torch.bfloat16, device=dev)
            if os.environ.get("RES"):
                kw.update(res1=torch.randn(M, N, device=dev).to(torch.bfloat16), ldr1=N, beta1=1.0)
        for _ in range(3):
            hip.gemm(A, W, out, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            hip.gemm(A, W, out, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        print(f"M={M} N={N} K={K} mode={mode} epi={epi}: {us:9.1f} us  {2.0 * M * N * K / us / 1e6:8.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
