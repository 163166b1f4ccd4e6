#!/usr/bin/env python
"""The temporal VAE decode of one request (14 frames, 72 x 128 latents -> 576 x 1024 uint8 frames) alone: wall time by HIP events,
and the kernel families inside it when run under `rocprofv3 --kernel-trace --stats` (tools/rocprof_summary.py on its database).

    python tools/vae_probe.py          (ITERS=3)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wiw_amd  # noqa: E402,F401
from wiw_amd import hip as H  # noqa: E402
from wiw_amd.vae import VAEHIP  # noqa: E402


def main():
    from wiw_amd import frontend as FE      # random-init VAE state dict (no checkpoint offline)
    dev = torch.device("cuda:0")
    hip = H.Hip(dev)
    vae = VAEHIP(FE.vae_random_state_dict(1), dev, hip=hip)
    lat = torch.randn(1, 14, 4, 72, 128, device=dev) * 0.5
    iters = int(os.environ.get("ITERS", "3"))
    out = vae.decode(lat)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        out = vae.decode(lat)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"VAE decode 14 x 576 x 1024: {ms:.1f} ms  ({97.2 / ms * 1e3:.0f} TFLOP/s of 97.2 TFLOP)  out {tuple(out.shape)} {out.dtype}")


if __name__ == "__main__":
    main()
