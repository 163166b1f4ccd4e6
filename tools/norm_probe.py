#!/usr/bin/env python
"""Streaming rates of the normalisation kernels at the UNet's shapes (HIP events, random data).
    python tools/norm_probe.py            # LayerNorm / GroupNorm at L0..L2 shapes of config A
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wiw_amd  # noqa: E402,F401
from wiw_amd import hip as H  # noqa: E402


def timeit(fn, iters=20):
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
    for (frames, S, C) in [(28, 9216, 320), (28, 2304, 640), (28, 576, 1280), (28, 9216, 640), (28, 9216, 960)]:
        M = frames * S
        x = torch.randn(M, C, device=dev).to(torch.bfloat16)
        g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
        out = torch.empty_like(x)
        sm = torch.empty_like(x)
        av = torch.randn(frames, C, device=dev)
        byts = M * C * 2
        us = timeit(lambda: hip.layernorm(x, M, C, g, b, out=out))
        print(f"layernorm        M={M} C={C}: {us:8.1f} us  {2 * byts / us / 1e6:6.2f} TB/s")
        us = timeit(lambda: hip.layernorm(x, M, C, g, b, addvec=av, addvec_ld=C, rows_per_vec=S, sum_out=sm, out=out))
        print(f"layernorm+sum    M={M} C={C}: {us:8.1f} us  {3 * byts / us / 1e6:6.2f} TB/s")
        stats = torch.zeros(frames * 64, device=dev)
        ab = torch.randn(frames * 2 * C, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        cnt = hip._gn_cnt_pool[0]
        ref = x.float().view(frames, S, 32, C // 32).transpose(1, 2).reshape(frames, 32, -1)
        ref_m, ref_v = ref.mean(-1), ref.var(-1, unbiased=False)
        for clip, rpu, label in ((False, S, "gn_stats        "), (True, 14 * S, "gn_stats (T*S)  ")):
            rpbs = [hip.gn_rows_per_block(rpu, clip)] + [int(v) for v in os.environ.get("RPB", "").split(",") if v]
            for rpb in rpbs:
                scratch = torch.empty(int(hip.lib.wiw_groupnorm_scratch_floats(M, rpu, rpb)), device=dev)
                assert int(hip.lib.wiw_groupnorm_counters(M, rpu, rpb)) <= cnt.numel()
                fn = lambda: hip.lib.wiw_groupnorm_stats(s, x.data_ptr(), C, None, 0, M, rpu, rpb, stats.data_ptr(), scratch.data_ptr(), cnt.data_ptr())
                us = timeit(fn)
                st = stats[: (M // rpu) * 64].view(-1, 32, 2)
                if not clip:
                    err = max(float((st[..., 0] - ref_m).abs().max()), float((st[..., 1] - ref_v).abs().max()))
                else:
                    rc = x.float().view(M // rpu, rpu, 32, C // 32).transpose(1, 2).reshape(M // rpu, 32, -1)
                    err = max(float((st[..., 0] - rc.mean(-1)).abs().max()), float((st[..., 1] - rc.var(-1, unbiased=False)).abs().max()))
                a0 = st.clone(); fn(); torch.cuda.synchronize()
                print(f"{label} M={M} C={C} rpb={rpb:4d}: {us:8.1f} us  {byts / us / 1e6:6.2f} TB/s   max|err| {err:.2e}  rerun identical {bool((a0 == st).all())}  counters zero {int(cnt.abs().sum()) == 0}")
        us = timeit(lambda: hip.lib.wiw_groupnorm_apply(s, x.data_ptr(), C, None, 0, M, S, ab.data_ptr(), 1, out.data_ptr()))
        print(f"gn_apply(silu)   M={M} C={C}: {us:8.1f} us  {2 * byts / us / 1e6:6.2f} TB/s")
        us = timeit(lambda: hip.groupnorm(x, C, None, 0, M, S, g, b, 1e-5, True, out=out))
        print(f"groupnorm (both) M={M} C={C}: {us:8.1f} us  {3 * byts / us / 1e6:6.2f} TB/s")


if __name__ == "__main__":
    main()
