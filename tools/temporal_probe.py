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
    B, T = 2, 14
    for S, heads in ((9216, 5), (2304, 10), (576, 20), (144, 20)):
        C = heads * 64
        M = B * T * S
        x = (torch.randn(M, C, device=dev) * 1.5 + 0.2).to(torch.bfloat16)
        wq, wk, wv = (torch.randn(C, C, device=dev) / math.sqrt(C) for _ in range(3))
        gamma, beta = 1 + 0.1 * torch.randn(C, device=dev), 0.1 * torch.randn(C, device=dev)
        wg, fold = pack_temporal_qkv(wq, wk, wv, gamma, beta)
        o = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
        us_f = timeit(lambda: hip.temporal_attn_block(x, wg, fold, o, C, B, T, S, heads, 1e-5, 0.125), iters)
        flops = 2.0 * M * 3 * C * C + 2.0 * B * S * heads * (2 * T * T * 64)
        line = f"S={S} C={C} M={M}: fused {us_f:8.1f} us  {flops / us_f / 1e6:7.1f} TFLOP/s"
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


if __name__ == "__main__":
    main()
