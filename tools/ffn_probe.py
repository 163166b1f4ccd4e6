#!/usr/bin/env python
"""Time the fused FeedForward kernel (csrc/ffn.hip) against the two-GEMM path it replaces (GEGLU up-projection + down
projection, wiw_gemm_bf16) and the LayerNorm pass in front, at the level-0 shape of the served UNet (M = 258 048, C = 320,
hidden 1280; random data, HIP events, interleaved rounds in ONE process).

    python tools/ffn_probe.py                 # M = 28 * 9216
    M=64512 ROUNDS=5 python tools/ffn_probe.py
Algorithmic FLOPs = 2 * M * (2560 * 320 + 320 * 1280); `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` of
ONLY_FUSED=1 runs measures the MFMA pipe of ffn_kernel alone."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wiw_amd  # noqa: E402,F401
from wiw_amd import hip as H  # noqa: E402
from wiw_amd.unet import pack_geglu  # noqa: E402

C, HID = 320, 1280


def timed(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    dev = torch.device("cuda:0")
    dt = torch.float16 if os.environ.get("DTYPE") == "fp16" else torch.bfloat16
    hip = H.Hip(dev, dt)
    M = int(os.environ.get("M", str(28 * 9216)))
    iters, rounds = int(os.environ.get("ITERS", "10")), int(os.environ.get("ROUNDS", "3"))
    only_fused = bool(os.environ.get("ONLY_FUSED"))
    x = (torch.randn(M, C, device=dev) * 1.5 + 0.2).to(dt)
    res = torch.randn(M, C, device=dev).to(dt)
    w0 = torch.randn(2 * HID, C, device=dev) / math.sqrt(C)
    b0 = torch.randn(2 * HID, device=dev) * 0.3
    w2 = H.TiledW((torch.randn(C, HID, device=dev) / math.sqrt(HID)).to(dt).contiguous())
    b2 = torch.randn(C, device=dev) * 0.3
    gamma, beta = 1 + 0.1 * torch.randn(C, device=dev), 0.1 * torch.randn(C, device=dev)
    rv = torch.randn(28, C, device=dev)
    w1f, b1f, _ = pack_geglu(w0, b0, H.FFN_CHUNK)
    w1f = H.TiledW(w1f.to(dt).contiguous())
    w1l, b1l, _ = pack_geglu(w0 * gamma[None, :], w0 @ beta + b0, H.FFN_CHUNK)
    w1l = H.TiledW(w1l.to(dt).contiguous())
    w1u, b1u, _ = pack_geglu(w0, b0)
    w1u = H.TiledW(w1u.to(dt).contiguous())
    out = torch.empty(M, C, dtype=dt, device=dev)
    a = torch.empty(M, C, dtype=dt, device=dev)
    g = torch.empty(M, HID, dtype=dt, device=dev)
    rpv = max(1, M // 28)
    kw = dict(rowvec=rv, rowvec_ld=C, rows_per_vec=rpv, res1=res, ldr1=C, beta1=1.0)

    def fused():
        hip.ffn_geglu(x, w1f, b1f, w2, b2, out, M, **kw)

    def fused_ln():
        hip.ffn_geglu(x, w1l, b1l, w2, b2, out, M, ln=True, **kw)

    def chain():
        hip.gemm(x, w1u, g, M=M, N=2 * HID, K=C, C1=C, bias=b1u, epilogue=H.EPI_GEGLU, n_out=HID)
        hip.gemm(g, w2, out, M=M, N=C, K=HID, C1=HID, bias=b2, **kw)

    def chain_ln():
        hip.layernorm(x, M, C, gamma, beta, out=a)
        hip.gemm(a, w1u, g, M=M, N=2 * HID, K=C, C1=C, bias=b1u, epilogue=H.EPI_GEGLU, n_out=HID)
        hip.gemm(g, w2, out, M=M, N=C, K=HID, C1=HID, bias=b2, **kw)

    variants = [("fused", fused), ("fused+LN", fused_ln)]
    if not only_fused:
        variants += [("gemm chain", chain), ("LN + gemm chain", chain_ln)]
    for _, fn in variants:
        for _ in range(2):
            fn()
    torch.cuda.synchronize()
    flops = 2.0 * M * 3 * HID * C
    best = {}
    for r in range(rounds):
        for name, fn in variants:
            us = timed(fn, iters)
            best[name] = min(best.get(name, 1e30), us)
            print(f"round {r} M={M} {name:16s} {us:9.1f} us  {flops / us / 1e6:7.1f} TFLOP/s", flush=True)
    for name, us in best.items():
        print(f"BEST M={M} {name:16s} {us:9.1f} us  {flops / us / 1e6:7.1f} TFLOP/s  ({flops / us / 1e6 / 2500 * 100:.1f} % of 2.5 PF/s)")


if __name__ == "__main__":
    main()
