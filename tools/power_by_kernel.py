#!/usr/bin/env python
"""Round 5: board power and shader clock per KERNEL — each served kernel launched back to back for a few seconds while a thread
samples rocm-smi.  Why: the denoising loop runs against the board's power management (2.04-2.09 GHz at ~1300 W,
profiles/r16j_power_clock.txt), so a kernel's cost is its energy; this prints rate, mean watts, mean clock and joules per TFLOP
(or per GB) of every family at its largest served shape.

    python tools/power_by_kernel.py            (WIW_LIB=... for a variant build;  SECONDS=4 per kernel)"""
import math
import os
import re
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wiw_amd  # noqa: E402,F401
from wiw_amd import hip as H  # noqa: E402
from wiw_amd.unet import pack_geglu  # noqa: E402


def smi():
    try:
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=10).stdout
        r = [x for x in out.splitlines() if x.startswith("card")][0]
        clocks = [int(x) for x in re.findall(r"\((\d+)Mhz\)", r)]
        return max(clocks[2:4]), float(r.split(",")[-1])
    except Exception:
        return None


def run(name, fn, work, unit, seconds):
    """work: TFLOP (unit 'TF') or GB (unit 'GB') per call"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    torch.cuda.synchronize()
    per = e0.elapsed_time(e1) / 5e3
    n = max(10, int(seconds / per))
    samples, stop = [], [False]

    def loop():
        while not stop[0]:
            s = smi()
            if s:
                samples.append(s)

    th = threading.Thread(target=loop)
    th.start()
    e0.record()
    done = 0
    while done < n:                        # keep the queue short: the sampler thread needs the GIL now and then
        for _ in range(min(50, n - done)):
            fn()
        done += 50
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop[0] = True
    th.join()
    sec = e0.elapsed_time(e1) / 1e3
    calls = min(done, ((n + 49) // 50) * 50)
    rate = work * calls / sec
    busy = samples[1:-1] if len(samples) > 4 else samples
    if busy:
        w = sum(s[1] for s in busy) / len(busy)
        f = sum(s[0] for s in busy) / len(busy)
        print(f"{name:46s} {per * 1e6:8.1f} us  {rate:8.1f} {unit}/s  {w:6.0f} W  {f:5.0f} MHz  {w / rate:6.3f} J/{unit}   ({len(busy)} samples)", flush=True)
    else:
        print(f"{name:46s} {per * 1e6:8.1f} us  {rate:8.1f} {unit}/s  (no samples)", flush=True)


def main():
    dev = torch.device("cuda:0")
    hip = H.Hip(dev)
    dt = hip.dtype
    sec = float(os.environ.get("SECONDS", "4"))
    g = torch.Generator(device=dev).manual_seed(1)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)

    def gemm_case(name, M, N, K, mode=0, C1=None, geglu=False, res=False, halo=False, Hh=0, Ww=0, T=14):
        C1 = C1 or K
        A = (rn(M, C1) * 0.5).to(dt)
        W = (rn(N, K) / math.sqrt(K)).to(dt)
        epi = 0
        if halo:
            from wiw_amd.unet import conv_k_halo32
            W = conv_k_halo32(W)
            epi |= H.K_HALO32
        Wt = H.TiledW(W.contiguous())
        n_out = N // 2 if geglu else N
        out = torch.empty(M, n_out, dtype=dt, device=dev)
        kw = dict(M=M, N=N, K=K, C1=C1, mode=mode, bias=rn(N))
        if mode:
            kw.update(H=Hh, Wd=Ww, T=T)
        if res:
            kw.update(res1=rn(M, N).to(dt), ldr1=N, beta1=1.0)
        if geglu:
            epi |= H.EPI_GEGLU
            kw.update(n_out=n_out)
        kw.update(epilogue=epi)
        run(name, lambda: hip.gemm(A, Wt, out, **kw), 2.0 * M * N * K / 1e12, "TF", sec)

    gemm_case("dense GEGLU up  64512 x 5120 x 640", 64512, 5120, 640, geglu=True)
    gemm_case("dense FF down   64512 x 640 x 2560 (+res)", 64512, 640, 2560, res=True)
    gemm_case("dense           258048 x 320 x 320 (+res)", 258048, 320, 320, res=True)
    gemm_case("conv_t3         64512 x 640 x 1920", 64512, 640, 1920, mode=H.A_CONV_T3, C1=640, Hh=36, Ww=64)
    try:
        gemm_case("conv3x3 halo    258048 x 320 x 5760", 258048, 320, 5760, mode=H.A_CONV3X3, C1=640, halo=True, Hh=72, Ww=128)
    except Exception as e:      # (helper names differ between builds: the halo case is optional)
        print("conv3x3 halo: skipped", type(e).__name__, e)
    gemm_case("conv3x3 per tap 4032 x 1280 x 11520", 4032, 1280, 11520, mode=H.A_CONV3X3, C1=1280, Hh=9, Ww=16)

    # spatial attention, S = 9216, 5 heads (level 0)
    frames, S, heads = 28, 9216, 5
    C = heads * 64
    M = frames * S
    q, k, v = rn(M, C), rn(M, C), rn(M, C)
    vt = v.to(dt).t().contiguous()
    qk = torch.cat([q, k], 1).to(dt).contiguous()
    qk_ps = torch.cat([q * (math.log2(math.e) / 8.0), k], 1).to(dt).contiguous()
    o = torch.empty(M, C, dtype=dt, device=dev)
    tf = 4.0 * frames * heads * S * S * 64 / 1e12
    run("spatial attention 32x32x16  S = 9216", lambda: hip.attn_spatial_ps(qk_ps, 2 * C, C, vt, M, o, C, frames, S, heads), tf, "TF", sec)
    run("spatial attention first form S = 9216", lambda: hip.attn_spatial(qk, 2 * C, C, vt, M, o, C, frames, S, heads, 0.125), tf, "TF", sec)

    # fused FeedForward, level 0
    Cf, HID = 320, 1280
    x = (rn(M, Cf) * 1.5 + 0.2).to(dt)
    w0, b0 = rn(2 * HID, Cf) / math.sqrt(Cf), rn(2 * HID) * 0.3
    w2 = H.TiledW((rn(Cf, HID) / math.sqrt(HID)).to(dt).contiguous())
    w1f, b1f, _ = pack_geglu(w0, b0, H.FFN_CHUNK)
    w1f = H.TiledW(w1f.to(dt).contiguous())
    of = torch.empty(M, Cf, dtype=dt, device=dev)
    b2f = rn(Cf)
    run("fused FeedForward 258048 x 320 (1280)", lambda: hip.ffn_geglu(x, w1f, b1f, w2, b2f, of, M, res1=x, ldr1=Cf, beta1=1.0, ln=True),
        2.0 * M * (2 * HID * Cf + Cf * HID) / 1e12, "TF", sec)

    # GroupNorm + SiLU, level 0
    gam, bet = 1 + 0.1 * rn(Cf), 0.1 * rn(Cf)
    run("GroupNorm + SiLU 258048 x 320", lambda: hip.groupnorm(x, Cf, None, 0, M, S, gam, bet, 1e-5, True, out=of), 3.0 * M * Cf * 2 / 1e9, "GB", sec)
    a1, a2 = torch.empty(1 << 28, dtype=torch.float32, device=dev), torch.empty(1 << 28, dtype=torch.float32, device=dev)
    run("device copy 1 GiB", lambda: a2.copy_(a1), 2.0 * (1 << 30) / 1e9, "GB", sec)


if __name__ == "__main__":
    main()
