#!/usr/bin/env python
"""Time the spatial / temporal attention kernels at the north-star level shapes (HIP events, random data)."""
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
    iters = int(os.environ.get("ITERS", "10"))
    for frames, S, heads in ((28, 9216, 5), (28, 2304, 10), (28, 576, 20), (28, 144, 20)):
        C = heads * 64
        M = frames * S
        qk = torch.randn(M, 2 * C, device=dev).to(torch.bfloat16)
        vt = torch.randn(C, M, device=dev).to(torch.bfloat16)
        o = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
        for _ in range(2):
            hip.attn_spatial(qk, 2 * C, C, vt, M, o, C, frames, S, heads, 0.125)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            hip.attn_spatial(qk, 2 * C, C, vt, M, o, C, frames, S, heads, 0.125)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        fl = 4.0 * frames * heads * S * S * 64
        print(f"attn_spatial frames={frames} S={S} heads={heads}: {us:9.1f} us  {fl / us / 1e6:8.1f} TFLOP/s", flush=True)
    for batch, T, S, heads in ((2, 14, 9216, 5), (2, 14, 2304, 10), (2, 14, 576, 20)):
        C = heads * 64
        M = batch * T * S
        qkv = torch.randn(M, 3 * C, device=dev).to(torch.bfloat16)
        o = torch.empty(M, C, dtype=torch.bfloat16, device=dev)
        for _ in range(2):
            hip.attn_temporal(qkv, 3 * C, o, C, batch, T, S, heads, 0.125)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            hip.attn_temporal(qkv, 3 * C, o, C, batch, T, S, heads, 0.125)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        gb = M * 4 * C * 2 / 1e9
        print(f"attn_temporal batch={batch} T={T} S={S} heads={heads}: {us:9.1f} us  {gb / us * 1e6 / 1e3:6.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
