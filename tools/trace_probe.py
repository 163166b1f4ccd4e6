#!/usr/bin/env python
"""Slot-boundary timeline of the 256x160 GEMM (debug build with -DWIW_TRACE, loaded through WIW_LIB): block 0 stamps
s_memtime at every barrier of its SECOND output tile for wave 0 (leading group) and wave 4 (lagging group).

    WIW_LIB=tools/ablate/libwiw_trace.so WIW_GEMM_TILE=big python tools/trace_probe.py M,N,K[,mode]
Prints, per K tile, the cycles of:  barrier wait | slot 0 (DMA issue + R k0) | wait | slot 1 (DMA + 20 MFMA) | wait |
slot 2 (DMA + R k1 + vmcnt) | wait | slot 3 (DMA + 20 MFMA)."""
import ctypes as C
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
    for spec in sys.argv[1:]:
        parts = [int(v) for v in spec.split(",")]
        M, N, K = parts[:3]
        mode = parts[3] if len(parts) > 3 else 0
        taps = {0: 1, 1: 9, 4: 3}[mode]
        C1 = K // taps
        A = torch.randn(M, C1, device=dev).to(torch.bfloat16)
        W = H.TiledW((torch.randn(N, K, device=dev) / math.sqrt(K)).to(torch.bfloat16))
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        kw = dict(M=M, N=N, K=K, C1=C1, mode=mode, bias=torch.randn(N, device=dev))
        if mode:
            kw.update(H=72 * 28 if mode == 1 else 1, Wd=128, T=14)
            if mode == 1:
                kw.update(H=M // 128, Wd=128)
        for _ in range(3):
            hip.gemm(A, W, out, **kw)
        torch.cuda.synchronize()
        buf = (C.c_longlong * (2 * 4096))()
        hip.lib.wiw_gemm_trace_read.argtypes = [C.c_void_p]
        assert hip.lib.wiw_gemm_trace_read(buf) == 0
        nk = K // 64
        for g, name in ((0, "wave 0 (lead)"), (1, "wave 4 (lag)")):
            t = [buf[g * 4096 + i] for i in range(2 + 8 * nk + 1)]
            print(f"--- M={M} N={N} K={K} mode={mode}: {name}; tile total {t[1 + 8 * nk] - t[0]} cycles, epilogue {t[2 + 8 * nk] - t[1 + 8 * nk]}")
            rows = []
            for kt in range(nk):
                b = 8 * kt
                prev = t[b] if kt == 0 else t[b]      # t[8kt] = end of previous slot 3 (or tile start)
                seg = [t[b + 1] - prev] + [t[b + j + 1] - t[b + j] for j in range(1, 8)]
                rows.append(seg)
            show = rows if nk <= 6 else rows[:3] + rows[nk // 2:nk // 2 + 2] + rows[-2:]
            for seg in show:
                print("   wait %5d | s0 %5d | wait %5d | s1 %5d | wait %5d | s2 %5d | wait %5d | s3 %5d" % tuple(seg))
            mid = rows[2:-1] if nk > 4 else rows
            avg = [sum(r[j] for r in mid) / len(mid) for j in range(8)]
            print("   mean (steady K tiles): " + " ".join(f"{a:6.0f}" for a in avg) + f"  = {sum(avg):.0f} cycles / K tile")


if __name__ == "__main__":
    main()
