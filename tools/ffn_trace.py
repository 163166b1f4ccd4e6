#!/usr/bin/env python
"""Slot timeline of the fused FeedForward kernel (csrc/ffn.hip) from a -DWIW_FFN_TRACE build: block 0 stamps s_memtime at
the slot boundaries of two steady-state chunks, wave 0 (H role) and wave 4 (Y role).

    python tools/build_variant.py trace ffn.hip:-DWIW_FFN_TRACE
    WIW_LIB=tools/ablate/libwiw_trace.so python tools/ffn_trace.py"""
import ctypes as C
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wiw_amd  # noqa: E402,F401
from wiw_amd import hip as H  # noqa: E402
from wiw_amd.unet import pack_geglu  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    hip = H.Hip(dev)
    M, Cn, HID = 28 * 9216, 320, 1280
    x = torch.randn(M, Cn, device=dev).to(torch.bfloat16)
    w0 = torch.randn(2 * HID, Cn, device=dev) / math.sqrt(Cn)
    w1, b1, _ = pack_geglu(w0, torch.randn(2 * HID, device=dev), H.FFN_CHUNK)
    w1 = H.TiledW(w1.to(torch.bfloat16).contiguous())
    w2 = H.TiledW((torch.randn(Cn, HID, device=dev) / math.sqrt(HID)).to(torch.bfloat16).contiguous())
    out = torch.empty(M, Cn, dtype=torch.bfloat16, device=dev)
    for _ in range(3):
        hip.ffn_geglu(x, w1, b1, w2, None, out, M, res1=x, ldr1=Cn, beta1=1.0)
    torch.cuda.synchronize()
    buf = (C.c_longlong * 128)()
    hip.lib.wiw_ffn_trace_read.argtypes = [C.c_void_p]
    assert hip.lib.wiw_ffn_trace_read(buf) == 0
    t = [[buf[r * 64 + i] for i in range(64)] for r in range(2)]
    hw = (C.c_uint * 8)()
    hip.lib.wiw_ffn_hwid_read.argtypes = [C.c_void_p]
    if hip.lib.wiw_ffn_hwid_read(hw) == 0:   # HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh[12] se[15:13]
        print("wave -> (simd, cu): " + "  ".join(f"w{w}: simd {(hw[w] >> 4) & 3} cu {(hw[w] >> 8) & 15}" for w in range(8)))
    for ch in range(2):
        h = t[0][ch * 32: ch * 32 + 14]
        y = t[1][ch * 32: ch * 32 + 18]
        print(f"--- chunk {ch}: H-wave (wave 0)")
        for kt in range(5):
            nxt = h[2 * kt + 2]
            print(f"   slot {kt}: barrier wait {h[2 * kt + 1] - h[2 * kt]:6d} | work (8 rd, 16 mfma, 8 rd, 16 mfma) {nxt - h[2 * kt + 1]:6d}")
        print(f"   slot 5: barrier wait {h[11] - h[10]:6d} | last 16 mfma {h[12] - h[11]:6d} | GEGLU + H write {h[13] - h[12]:6d}")
        g = t[0][ch * 32 + 20: ch * 32 + 22]
        print(f"   GEGLU by pass of 16 gate values (from 'last k-step issued'): {g[0] - h[12]:6d} {g[1] - g[0]:6d}")
        print(f"   chunk total (slot-0 stamp to GEGLU done): {h[13] - h[0]:6d} cycles")
        print(f"--- chunk {ch}: Y-wave (wave 4)")
        if os.environ.get("FREEK"):   # -DFFN_FREE_K=1 build: two barriers per chunk
            y = t[1][ch * 32: ch * 32 + 19]
            nxt = t[1][(ch + 1) * 32] if ch == 0 else None
            print(f"   barrier A wait {y[1] - y[0]:6d} | W2 + bias DMA issue {y[2] - y[1]:6d} | vmcnt wait {y[15] - y[2]:6d} | barrier B wait {y[16] - y[15]:6d}"
                  f" | W1 bulk issue + phase 2 {y[17] - y[16]:6d} | epilogue (if any) {y[18] - y[17]:6d}"
                  + (f" | W1 vmcnt wait {nxt - y[18]:6d}" if nxt is not None else ""))
            continue
        for kt in range(5):
            nxt = y[3 * kt + 3] if kt < 4 else y[15]
            print(f"   slot {kt}: barrier wait {y[3 * kt + 1] - y[3 * kt]:6d} | DMA issue {y[3 * kt + 2] - y[3 * kt + 1]:6d} | vmcnt wait {nxt - y[3 * kt + 2]:6d}")
        print(f"   slot 5: barrier wait {y[16] - y[15]:6d} | phase 2 (80 mfma) {y[17] - y[16]:6d}")
        print(f"   chunk total: {y[17] - y[0]:6d} cycles")
    print("chunk-to-chunk period (H-wave):", t[0][32] - t[0][0], "cycles")


if __name__ == "__main__":
    main()
