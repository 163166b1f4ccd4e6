#!/usr/bin/env python
"""Round 5: the 32x32x16 fused FeedForward (wiw_ffn32_geglu, csrc/ffn32.hip) against the fp32 oracle and against ffn.hip."""
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wiw_amd  # noqa: E402,F401
from wiw_amd import hip as H  # noqa: E402
from wiw_amd.unet import pack_ffn32, pack_geglu  # noqa: E402

C, HID = 320, 1280


def main():
    dev = torch.device("cuda:0")
    hip = H.Hip(dev)
    dt = hip.dtype
    g = torch.Generator().manual_seed(0)
    w0 = torch.randn(2 * HID, C, generator=g) / math.sqrt(C)
    b0 = torch.randn(2 * HID, generator=g) * 0.3
    w2 = torch.randn(C, HID, generator=g) / math.sqrt(HID)
    b2 = torch.randn(C, generator=g) * 0.3
    W1s, b1s, W2s = pack_ffn32(w0.to(dev), b0.to(dev), w2.to(dev), dt)
    w1p, b1p, _ = pack_geglu(w0, b0, H.FFN_CHUNK)
    W1, b1, W2 = H.TiledW(w1p.to(dt).to(dev).contiguous()), b1p.to(dev).contiguous(), H.TiledW(w2.to(dt).to(dev).contiguous())
    b2d = b2.to(dev)
    for M in [int(v) for v in os.environ.get("MS", "128,1000,33000,258048").split(",")]:
        x = (torch.randn(M, C, generator=g)).to(dt)
        r1 = torch.randn(M, C, generator=g).to(dt)
        xd, r1d = x.to(dev), r1.to(dev)
        o1 = torch.empty(M, C, dtype=dt, device=dev)
        o2 = torch.full((M, C), float("nan"), dtype=dt, device=dev)
        o3 = torch.empty_like(o2)
        kw = dict(res1=r1d, ldr1=C, beta1=1.0)
        hip.ffn_geglu(xd, W1, b1, W2, b2d, o1, M, **kw)
        hip.ffn32_geglu(xd, W1s, b1s, W2s, b2d, o2, M, **kw)
        hip.ffn32_geglu(xd, W1s, b1s, W2s, b2d, o3, M, **kw)
        torch.cuda.synchronize()
        if M <= 40000:
            xf = x.float()
            proj = xf @ w0.to(dt).float().t() + b0
            hid, gate = proj.chunk(2, dim=-1)
            h = (hid * F.gelu(gate)).to(dt).float()
            ref = h @ w2.to(dt).float().t() + b2 + r1.float()
            for name, o in (("v1", o1), ("ffn32", o2)):
                d = o.float().cpu() - ref
                print(f"   M={M} {name:6s} vs fp32 oracle: max_rel {float(d.abs().max() / ref.abs().max()):.3e} rms_rel "
                      f"{float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.3e} nan {int(torch.isnan(o.float()).sum())}", flush=True)
        print(f"   M={M} ffn32 repeatable: {bool(torch.equal(o2, o3))}   max |v1 - ffn32| {float((o1.float() - o2.float()).abs().max()):.3e}", flush=True)
        for name, fn in (("v1", lambda: hip.ffn_geglu(xd, W1, b1, W2, b2d, o1, M, **kw)),
                         ("ffn32", lambda: hip.ffn32_geglu(xd, W1s, b1s, W2s, b2d, o2, M, **kw))):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 5
            fl = 2.0 * M * (2 * HID * C + C * HID)
            print(f"ffn {name:6s} M={M}: {us:9.1f} us  {fl / us / 1e6:8.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
