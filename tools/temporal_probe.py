#!/usr/bin/env python
"""Time the fused temporal block (temporal.hip) against the unfused chain (LayerNorm kernel -> QKV GEMM ->
wiw_attn_temporal_bf16) at the three levels of the served UNet (random data, HIP events).

    python tools/temporal_probe.py            # B=1 with CFG: batch 2, T = 14
The fused kernel's algorithmic FLOPs = 2*M*3C*C (projection) + 4*16*16*64*heads*sites*... (attention, counted at the
real T); both are reported so `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` of THIS process measures the
MFMA pipe of temporal_block_kernel alone."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wiw_amd  # noqa: E402,F401
from wiw_amd import hip as H  # noqa: E402
from wiw_amd.unet import pack_temporal_qkv  # noqa: E402


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    dev = torch.device("cuda:0")
    hip = H.Hip(dev)
    iters = int(os.environ.get("ITERS", "20"))
    only_fused = bool(os.environ.get("ONLY_FUSED"))
    shapes = [(2, 14, 9216, 5), (2, 14, 2304, 10), (2, 14, 576, 20), (2, 14, 144, 20)]
    if os.environ.get("SHAPES"):   # "B,T,S,heads;..."  e.g. 1152,14,16,5: the L0 work with all frames of a site tile in one 2-MiB page
        shapes = [tuple(int(v) for v in sp.split(",")) for sp in os.environ["SHAPES"].split(";")]
    for B, T, S, heads in shapes:
        C = heads * 64
        M = B * T * S
        x = (torch.randn(M, C, device=dev) * 1.5 + 0.2).to(torch.bfloat16)
        wq, wk, wv = (torch.randn(C, C, device=dev) / math.sqrt(C) for _ in range(3))
        gamma, beta = 1 + 0.1 * torch.randn(C, device=dev), 0.1 * torch.randn(C, device=dev)
        wg, fold = pack_temporal_qkv(wq, wk, wv, gamma, beta)
        o = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
        us_f = timeit(lambda: hip.temporal_attn_block(x, wg, fold, o, C, B, T, S, heads, 1e-5, 0.125), iters)
        flops = 2.0 * M * 3 * C * C + 2.0 * B * S * heads * (2 * T * T * 64)
        line = f"B={B} T={T} S={S} C={C} M={M}: fused {us_f:8.1f} us  {flops / us_f / 1e6:7.1f} TFLOP/s"
        if not only_fused:
            wqkv = torch.cat([wq, wk, wv]).to(torch.bfloat16).contiguous()
            a = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
            qkv = torch.empty(M, 3 * C, dtype=torch.bfloat16, device=dev)
            o2 = torch.empty_like(o)

            def chain():
                hip.layernorm(x, M, C, gamma, beta, out=a)
                hip.gemm(a, wqkv, qkv, M=M, N=3 * C, K=C, C1=C)
                hip.attn_temporal(qkv, 3 * C, o2, C, B, T, S, heads, 0.125)

            us_c = timeit(chain, iters)
            line += f" | unfused chain {us_c:8.1f} us ({us_c / us_f:.2f}x)"
        print(line, flush=True)
        if hasattr(hip.lib, "wiw_temporal_rtrace_read") and C <= 320 and not os.environ.get("WIW_TEMPORAL_RING"):
            import ctypes as Ct
            rb = (Ct.c_longlong * 128)()
            hip.lib.wiw_temporal_rtrace_read.argtypes = [Ct.c_void_p]
            assert hip.lib.wiw_temporal_rtrace_read(rb) == 0
            t0 = min(rb[w * 16] for w in range(8))
            nk = C // 64
            for w in range(8):
                t = [rb[w * 16 + i] for i in range(16)]
                kt = " ".join(f"[ld+wait {t[2 + 2 * k] - t[1 + 2 * k]} mma {(t[3 + 2 * k] if k + 1 < nk else t[11]) - t[2 + 2 * k]}]" for k in range(nk))
                print(f"   wave {w}: item starts at +{t[0] - t0}, K tiles {kt} | epilogue {t[12] - t[11]} | item {t[12] - t[0]}")


if __name__ == "__main__":
    main()
