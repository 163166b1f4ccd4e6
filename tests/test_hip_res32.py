"""The fp32 RESIDUAL STREAM (ABI 11, `UNetHIP(residual_fp32=True)`) and the tolerance north_star states.  GPU only.

The reference keeps its latents and the Euler update in fp32 (scheduling_euler_discrete.py:635,673), evaluates the UNet in
fp16 (eval_inference.py:294) and, being PyTorch, rounds every tensor it writes to that type.  A 16-bit library that rounds
the residual stream after every block is 1.2e-3 (fp16) away from the reference's fp32 output on the served architecture;
`oracle/precision_study.py` says the stream in fp32 (every MFMA operand still 16-bit) brings that under north_star's 1e-3.
This file gates it:

  * kernels of the mode against fp32 torch on identical inputs: GEMM / conv epilogue with fp32 residuals and fp32 or 16-bit
    output (fragment-layout epilogue: ONE rounding, none for fp32 out), split-K, GroupNorm / LayerNorm on fp32 inputs, the
    rounded raw copy, the cast;
  * the served-width network (tests/golden/unet_full_16x32.npz, produced by the reference):
        fp16 + fp32 stream   vs the reference's fp32 output          rms <= 1.0e-3   (north_star)
        fp16 (both streams)  vs the reference on the SAME fp16-rounded weights, fp32 arithmetic   rms <= 1.0e-3
        bf16 (both streams)  vs the reference on the SAME bf16-rounded weights: 1.2 x the measured value
  * the reference's own 25-step trajectory at served width (tests/golden/pipeline_full_16x32.npz,
    `StableVideoDiffusionPipeline.__call__`, pipeline_stable_video_diffusion.py:383-638) and BASELINE config 0's workload
    (256x256x8, 10 steps: pipeline_config0_32x32.npz) — relative latent error of the HIP loop, every build;
  * the identities of the 16-bit path hold in this mode too: candidate independence and graph replay, bit for bit.
Gates are ~1.25 x the deterministic measurement of the round (printed in profiles/r08*_gpu_suite.log), never looser than the
statement they stand for.
"""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DTYPES = {"fp16": torch.float16, "bf16": torch.bfloat16}


def rel(a, b):
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().float().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    a, b = a.astype(np.float64), b.astype(np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30)), float(
        np.sqrt(((a - b) ** 2).mean()) / (np.sqrt((b ** 2).mean()) + 1e-30))


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


_HIPS = {}


def get_hip(name):
    import wiw_amd  # noqa: F401
    from wiw_amd.hip import Hip

    if name not in _HIPS:
        _HIPS[name] = Hip(torch.device(DEV), DTYPES[name])
    return _HIPS[name]


def nhwc(x):
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1])


# ----------------------------------------------------------------------------------------------
# kernels
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["fp16", "bf16"])
@pytest.mark.parametrize("M,N,K", [(4096 + 40, 320, 640), (300, 1280, 320), (33000, 640, 1280),
                                   (51200 + 40, 320, 640), (25700, 640, 320)])   # the last three: the 256x320 tile's F32E instantiation
def test_gemm_fp32_residuals(name, M, N, K):
    """out = alpha (A W^T + bias + vec[row group]) + beta1 res1 + beta2 res2 with fp32 residuals: fp32 output (exact up to
    the accumulation order) and 16-bit output (one rounding of the fp32 sum)."""
    from wiw_amd.hip import EPI_OUT_F32, TiledW

    hip, dt = get_hip(name), DTYPES[name]
    A = rnd(M, K, seed=1).to(dt)
    W = (rnd(N, K, seed=2) / math.sqrt(K)).to(dt)
    bias, vec = rnd(N, seed=3), rnd(-(-M // 64), N, seed=4)
    r1, r2 = rnd(M, N, seed=5) * 3.0, rnd(M, N, seed=6)
    alpha, b1, b2 = 0.7, 1.0, 0.3
    ref = alpha * (A.float() @ W.float().t() + bias + vec.repeat_interleave(64, 0)[:M]) + b1 * r1 + b2 * r2
    Ad, Wd = A.to(DEV), TiledW(W.to(DEV))
    args = dict(M=M, N=N, K=K, C1=K, bias=bias.to(DEV), rowvec=vec.to(DEV), rowvec_ld=N, rows_per_vec=64, alpha=alpha,
                res1=r1.to(DEV), ldr1=N, beta1=b1, res2=r2.to(DEV), ldr2=N, beta2=b2)
    out32 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    hip.gemm(Ad, Wd, out32, epilogue=EPI_OUT_F32, **args)
    mx, rms = rel(out32, ref)
    print(f"[res32 {name}] gemm {M}x{N}x{K} fp32 residuals -> fp32: max_rel={mx:.2e} rms_rel={rms:.2e}")
    assert torch.isfinite(out32).all() and mx <= 2e-5 and rms <= 3e-6
    out16 = torch.empty(M, N, dtype=dt, device=DEV)
    hip.gemm(Ad, Wd, out16, **args)
    assert torch.equal(out16.cpu(), out32.cpu().to(dt)), "16-bit output must be the ONE rounding of the fp32 result"
    # a 16-bit res1 next to an fp32 res2 (mixed), against the same formula on the rounded res1
    r1h = r1.to(dt)
    hip.gemm(Ad, Wd, out32, epilogue=EPI_OUT_F32, **dict(args, res1=r1h.to(DEV)))
    mx, rms = rel(out32, ref - b1 * r1 + b1 * r1h.float())
    assert mx <= 2e-5 and rms <= 3e-6


@pytest.mark.parametrize("name", ["fp16", "bf16"])
def test_conv_fp32_stream_epilogues(name):
    """The two convolutions of a SpatioTemporalResBlock that write the stream: conv2 (+ fused 1x1 shortcut over the rounded
    block input) -> fp32, temporal conv2 + AlphaBlender with an fp32 residual -> fp32; plus split-K with an fp32 residual."""
    from wiw_amd.hip import A_CONV3X3, A_CONV_T3, EPI_OUT_F32, TiledW

    hip, dt = get_hip(name), DTYPES[name]
    Nf, Cin, Cout, H, W = 4, 128, 64, 8, 16
    x = rnd(Nf, Cout, H, W, seed=1).to(dt)
    raw = rnd(Nf, Cin, H, W, seed=2).to(dt)
    w2 = (rnd(Cout, Cout, 3, 3, seed=3) / math.sqrt(9 * Cout)).to(dt)
    wsc = (rnd(Cout, Cin, 1, 1, seed=4) / math.sqrt(Cin)).to(dt)
    b = rnd(Cout, seed=5)
    ref = F.conv2d(x.float(), w2.float(), b, padding=1) + F.conv2d(raw.float(), wsc.float())
    Wcat = torch.cat([w2.float().permute(0, 2, 3, 1).reshape(Cout, -1), wsc.float()[:, :, 0, 0]], dim=1).to(dt)
    M = Nf * H * W
    out = torch.empty(M, Cout, dtype=torch.float32, device=DEV)
    hip.gemm(nhwc(x).contiguous().to(DEV), TiledW(Wcat.to(DEV)), out, M=M, N=Cout, K=9 * Cout + Cin, C1=Cout, mode=A_CONV3X3,
             H=H, Wd=W, A2=nhwc(raw).contiguous().to(DEV), C2=Cin, bias=b.to(DEV), epilogue=EPI_OUT_F32)
    mx, rms = rel(out, nhwc(ref))
    print(f"[res32 {name}] conv3x3 + shortcut -> fp32: max_rel={mx:.2e} rms_rel={rms:.2e}")
    assert mx <= 2e-5 and rms <= 3e-6
    # temporal conv + blend: xs + (1 - a) (conv_t(h) + b), xs fp32
    T, a = 4, 0.37
    wt = (rnd(Cout, Cout, 3, 1, 1, seed=6) / math.sqrt(3 * Cout)).to(dt)
    xs = rnd(Nf, Cout, H, W, seed=7) * 2.0
    x5 = x.float().reshape(1, T, Cout, H, W).permute(0, 2, 1, 3, 4)
    ct = F.conv3d(x5, wt.float(), b, padding=(1, 0, 0)).permute(0, 2, 1, 3, 4).reshape(Nf, Cout, H, W)
    ref = xs + (1.0 - a) * ct
    Wt = wt.float()[:, :, :, 0, 0].permute(0, 2, 1).reshape(Cout, -1).to(dt)
    hip.gemm(nhwc(x).contiguous().to(DEV), TiledW(Wt.to(DEV)), out, M=M, N=Cout, K=3 * Cout, C1=Cout, mode=A_CONV_T3, H=H, Wd=W,
             T=T, bias=b.to(DEV), alpha=1.0 - a, res1=nhwc(xs).contiguous().to(DEV), ldr1=Cout, beta1=1.0, epilogue=EPI_OUT_F32)
    mx, rms = rel(out, nhwc(ref))
    print(f"[res32 {name}] temporal conv + AlphaBlender, fp32 residual -> fp32: max_rel={mx:.2e} rms_rel={rms:.2e}")
    assert mx <= 2e-5 and rms <= 3e-6
    # split-K (the 1280-channel level): fp32 residual + fp32 out through splitk_reduce_kernel
    C = 320
    xk = rnd(2, C, 8, 8, seed=8).to(dt)
    wk = (rnd(C, C, 3, 3, seed=9) / math.sqrt(9 * C)).to(dt)
    rk = rnd(2, C, 8, 8, seed=10)
    ref = F.conv2d(xk.float(), wk.float(), None, padding=1) + rk
    outk = torch.empty(128, C, dtype=torch.float32, device=DEV)
    hip.gemm(nhwc(xk).contiguous().to(DEV), TiledW(wk.float().permute(0, 2, 3, 1).reshape(C, -1).to(dt).to(DEV)), outk, M=128, N=C,
             K=9 * C, C1=C, mode=A_CONV3X3, H=8, Wd=8, res1=nhwc(rk).contiguous().to(DEV), ldr1=C, beta1=1.0, splitk=3,
             epilogue=EPI_OUT_F32)
    mx, rms = rel(outk, nhwc(ref))
    assert mx <= 2e-5 and rms <= 3e-6


@pytest.mark.parametrize("name", ["fp16", "bf16"])
def test_conv_fp32_stream_on_the_256x320_tile(name):
    """The fp32-stream epilogue of `gemm_huge_kernel<…, F32E>` (round 4, second half) at sizes the 256x320 tile takes: halo-staged
    (plain, shortcut segment, upsample) and per-tap 3x3 convolutions and the temporal convolution, fp32 output, fp32 residual,
    per-frame vector — against fp32 torch on the operands the kernel sees."""
    from wiw_amd.hip import A_CONV3X3, A_CONV3X3_UP, A_CONV_T3, EPI_OUT_F32, K_CMAJOR, K_HALO32, TiledW
    from wiw_amd.unet import conv_k_cmajor, conv_k_halo32

    hip, dt = get_hip(name), DTYPES[name]
    B, T, Cin, Cout, H, W = 2, 14, 128, 320, 36, 64
    n = B * T
    M = n * H * W
    x = rnd(n, Cout, H, W, seed=1).to(dt)
    raw = rnd(n, Cin, H, W, seed=2).to(dt)
    res = rnd(n, Cout, H, W, seed=3) * 2.0
    vec = rnd(n, Cout, seed=4)
    w2 = (rnd(Cout, Cout, 3, 3, seed=5) / math.sqrt(9 * Cout)).to(dt)
    wsc = (rnd(Cout, Cin, seed=6) / math.sqrt(Cin)).to(dt)
    b = rnd(Cout, seed=7)
    w_tap = w2.permute(0, 2, 3, 1).reshape(Cout, -1)
    xd, resd = nhwc(x).contiguous().to(DEV), nhwc(res).contiguous().to(DEV)
    conv = F.conv2d(x.float(), w2.float(), b, padding=1)
    ref_a = conv + F.conv2d(raw.float(), wsc.float()[:, :, None, None])
    ref_b = 0.75 * (conv + vec[:, :, None, None]) + res
    for (label, wk, epi) in (("halo", conv_k_halo32(w_tap), K_HALO32), ("per-tap", conv_k_cmajor(w_tap, 9), K_CMAJOR)):
        out = torch.empty(M, Cout, dtype=torch.float32, device=DEV)
        hip.gemm(xd, TiledW(torch.cat([wk, wsc], 1).contiguous().to(DEV)), out, M=M, N=Cout, K=9 * Cout + Cin, C1=Cout,
                 mode=A_CONV3X3, H=H, Wd=W, A2=nhwc(raw).contiguous().to(DEV), C2=Cin, bias=b.to(DEV), epilogue=EPI_OUT_F32 | epi)
        mx, rms = rel(out, nhwc(ref_a))
        print(f"[res32 {name}] 256x320 tile, conv3x3 + shortcut ({label}) -> fp32: max_rel={mx:.2e} rms_rel={rms:.2e}")
        assert mx <= 2e-5 and rms <= 3e-6
        hip.gemm(xd, TiledW(wk.contiguous().to(DEV)), out, M=M, N=Cout, K=9 * Cout, C1=Cout, mode=A_CONV3X3, H=H, Wd=W,
                 bias=b.to(DEV), rowvec=vec.to(DEV), rowvec_ld=Cout, rows_per_vec=H * W, res1=resd, ldr1=Cout, beta1=1.0, alpha=0.75,
                 epilogue=EPI_OUT_F32 | epi)
        mx, rms = rel(out, nhwc(ref_b))
        print(f"[res32 {name}] 256x320 tile, conv3x3 + vector + fp32 residual ({label}) -> fp32: max_rel={mx:.2e} rms_rel={rms:.2e}")
        assert mx <= 2e-5 and rms <= 3e-6
        out16 = torch.empty(M, Cout, dtype=dt, device=DEV)      # fp32 residual, 16-bit output: ONE rounding of the same sum
        hip.gemm(xd, TiledW(wk.contiguous().to(DEV)), out16, M=M, N=Cout, K=9 * Cout, C1=Cout, mode=A_CONV3X3, H=H, Wd=W,
                 bias=b.to(DEV), rowvec=vec.to(DEV), rowvec_ld=Cout, rows_per_vec=H * W, res1=resd, ldr1=Cout, beta1=1.0, alpha=0.75,
                 epilogue=epi)
        assert torch.equal(out16.cpu(), out.cpu().to(dt))
    xs = rnd(n, Cout, H // 2, W // 2, seed=8).to(dt)
    ref_u = F.conv2d(F.interpolate(xs.float(), scale_factor=2.0, mode="nearest"), w2.float(), b, padding=1)
    for (label, wk, epi) in (("halo", conv_k_halo32(w_tap), K_HALO32), ("per-tap", conv_k_cmajor(w_tap, 9), K_CMAJOR)):
        out = torch.empty(M, Cout, dtype=torch.float32, device=DEV)
        hip.gemm(nhwc(xs).contiguous().to(DEV), TiledW(wk.contiguous().to(DEV)), out, M=M, N=Cout, K=9 * Cout, C1=Cout,
                 mode=A_CONV3X3_UP, H=H, Wd=W, bias=b.to(DEV), epilogue=EPI_OUT_F32 | epi)
        mx, rms = rel(out, nhwc(ref_u))
        print(f"[res32 {name}] 256x320 tile, upsample + conv3x3 ({label}) -> fp32: max_rel={mx:.2e} rms_rel={rms:.2e}")
        assert mx <= 2e-5 and rms <= 3e-6
    a = 0.3
    wt = (rnd(Cout, Cout, 3, 1, 1, seed=9) / math.sqrt(3 * Cout)).to(dt)
    x5 = x.float().reshape(B, T, Cout, H, W).permute(0, 2, 1, 3, 4)
    ct = F.conv3d(x5, wt.float(), b, padding=(1, 0, 0)).permute(0, 2, 1, 3, 4).reshape(n, Cout, H, W)
    out = torch.empty(M, Cout, dtype=torch.float32, device=DEV)
    Wt = conv_k_cmajor(wt[:, :, :, 0, 0].permute(0, 2, 1).reshape(Cout, -1), 3)
    hip.gemm(xd, TiledW(Wt.contiguous().to(DEV)), out, M=M, N=Cout, K=3 * Cout, C1=Cout, mode=A_CONV_T3, H=H, Wd=W, T=T,
             bias=b.to(DEV), alpha=1.0 - a, res1=resd, ldr1=Cout, beta1=1.0, epilogue=EPI_OUT_F32 | K_CMAJOR)
    mx, rms = rel(out, nhwc(res + (1.0 - a) * ct))
    print(f"[res32 {name}] 256x320 tile, temporal conv + AlphaBlender, fp32 residual -> fp32: max_rel={mx:.2e} rms_rel={rms:.2e}")
    assert mx <= 2e-5 and rms <= 3e-6


@pytest.mark.parametrize("name", ["fp16", "bf16"])
def test_norms_on_fp32_inputs(name):
    hip, dt = get_hip(name), DTYPES[name]
    # GroupNorm(32)+SiLU over the concat of two fp32 tensors, per frame and per clip; the rounded raw copy
    Nf, C1, C2, H, W = 4, 64, 64, 8, 16
    x1, x2 = rnd(Nf, C1, H, W, seed=1) * 2 + 0.5, rnd(Nf, C2, H, W, seed=2)
    g, b = 1 + 0.1 * rnd(C1 + C2, seed=3), 0.1 * rnd(C1 + C2, seed=4)
    cat = torch.cat([x1, x2], 1)
    M, S = Nf * H * W, H * W
    for rpu, clip in ((S, False), (Nf * S, True)):
        if clip:
            ref = F.silu(F.group_norm(cat.reshape(1, Nf, C1 + C2, H, W).permute(0, 2, 1, 3, 4), 32, g, b, 1e-6))
            ref = ref.permute(0, 2, 1, 3, 4).reshape(Nf, C1 + C2, H, W)
        else:
            ref = F.silu(F.group_norm(cat, 32, g, b, 1e-6))
        raw = torch.empty(M, C1 + C2, dtype=dt, device=DEV)
        out = hip.groupnorm(nhwc(x1).contiguous().to(DEV), C1, nhwc(x2).contiguous().to(DEV), C2, M, rpu, g.to(DEV), b.to(DEV),
                            1e-6, True, clip=clip, raw16=raw)
        mx, rms = rel(out, nhwc(ref))
        print(f"[res32 {name}] GroupNorm+SiLU on fp32 inputs (clip={clip}): max_rel={mx:.2e} rms_rel={rms:.2e}")
        eps16 = 2.0 ** (-11 if dt == torch.float16 else -8)
        assert mx <= 1.5 * eps16 and rms <= 0.6 * eps16
        assert torch.equal(raw.cpu(), nhwc(cat).to(dt)), "raw16 must be the rounded input, concatenated"
    # LayerNorm on fp32 rows at the three widths
    for C in (320, 640, 1280):
        x = rnd(1000 + 7, C, seed=C) * 1.5 + 0.3
        g, b = 1 + 0.1 * rnd(C, seed=1), 0.1 * rnd(C, seed=2)
        out = hip.layernorm(x.to(DEV), x.shape[0], C, g.to(DEV), b.to(DEV))
        ref = F.layer_norm(x, (C,), g, b, 1e-5)
        mx, rms = rel(out, ref)
        assert out.dtype == dt and mx <= 1.5 * eps16 and rms <= 0.6 * eps16
    # cast
    x = rnd(4096 + 8, seed=11) * 100
    assert torch.equal(hip.cast16(x.to(DEV)).cpu(), x.to(dt))


# ----------------------------------------------------------------------------------------------
# the served architecture
# ----------------------------------------------------------------------------------------------
_UNETS = {}


def label32(res32):
    """residual_fp32 of UNetHIP: False | True (the block-level stream in fp32, the mode since round 6) | "full" (rounds 4-5: also
    the hidden stream inside the transformer blocks)."""
    return {False: "", True: " + fp32 residual stream", "full": " + fp32 residual stream (full: also inside the transformer blocks)"}[res32]


def full_unet(name, res32):
    import wiw_amd  # noqa: F401
    from wiw_amd.config import UNetConfig
    from wiw_amd.unet import UNetHIP
    from wiw_amd.weights import random_state_dict

    key = (name, res32)
    if key not in _UNETS:
        cfg = UNetConfig()
        sd = {k: torch.from_numpy(v) for k, v in random_state_dict(cfg, 4).items()}      # the fixtures' weight seed
        _UNETS[key] = UNetHIP(cfg, sd, DEV, hip=get_hip(name), residual_fp32=res32)
        assert _UNETS[key].res32_any == bool(res32) and _UNETS[key].res32_tr == (res32 == "full")
    return _UNETS[key]


def run_unet(unet, g):
    return unet(torch.from_numpy(g["sample"]), float(g["timestep"]), torch.from_numpy(g["ehs"]),
                torch.from_numpy(g["added_time_ids"]), torch.from_numpy(g["action_ids"])).cpu().numpy()


# (build, fp32 stream) -> gates: rms vs the reference's fp32 output, rms vs the reference on the SAME rounded weights
# measured in round 4 (profiles/r08b_res32_tests.log; deterministic kernels):   vs fp32      vs same weights
#   fp16 + fp32 stream   8.64e-4   8.77e-4        fp16   1.210e-3   1.232e-3
#   bf16 + fp32 stream   7.09e-3   7.08e-3        bf16   9.85e-3    9.78e-3
FORWARD_GATES = {
    ("fp16", True): (1.0e-3, 1.0e-3),      # north_star's tolerance, both against fp32 weights and on the arithmetic alone (block-level
                                           # stream, round 6: 9.47e-4 / 9.68e-4)
    ("fp16", "full"): (1.0e-3, 1.0e-3),    # the full fp32 stream of rounds 4-5: 8.61e-4 / 8.77e-4
    ("fp16", False): (1.45e-3, 1.48e-3),   # 1.2 x measured (the reference's own fp16 run is 1.50e-3 from its fp32 run)
    ("bf16", True): (8.5e-3, 8.5e-3),      # 1.2 x measured
    ("bf16", False): (1.18e-2, 1.17e-2),   # 1.2 x measured (the reference's own bf16 run: 1.27e-2)
}


@pytest.mark.parametrize("name,res32", list(FORWARD_GATES))
def test_unet_full_width_tolerance(name, res32, golden):
    g = golden("unet_full_16x32.npz")
    assert int(g["weight_seed"]) == 4
    out = run_unet(full_unet(name, res32), g)
    mx, rms = rel(out, g["out"])
    mx_w, rms_w = rel(out, g[f"out_ref_{name}_weights_fp32_math"])
    mx_own, rms_own = rel(g[f"out_ref_{name}"], g["out"])
    floor = rel(g[f"out_ref_{name}_weights_fp32_math"], g["out"])[1]
    print(f"[tolerance] FULL-WIDTH unet 16x32x14, {name}{label32(res32)}: vs reference fp32 "
          f"rms={rms:.3e} max={mx:.3e} | vs reference on the SAME {name}-rounded weights (fp32 math) rms={rms_w:.3e} max={mx_w:.3e} | "
          f"reference's own {name} run {rms_own:.3e} | weight-rounding floor {floor:.3e}")
    assert np.isfinite(out).all()
    gate, gate_w = FORWARD_GATES[(name, res32)]
    assert rms <= gate, f"{name} res32={res32}: rms {rms:.3e} vs the reference's fp32 output exceeds {gate:.2e}"
    assert rms_w <= gate_w, f"{name} res32={res32}: rms {rms_w:.3e} vs the same-weights reference exceeds {gate_w:.2e}"
    assert rms <= rms_own          # never worse than the reference's own 16-bit run


def test_res32_candidate_independence_and_graph(golden):
    """The identities of the 16-bit path in the fp32-stream mode: B = 2 rows equal their B = 1 runs bit for bit; a captured
    hipGraph of the forward replays the eager bytes."""
    from wiw_amd.pipeline import SVDDenoiser

    g = golden("unet_full_16x32.npz")
    unet = full_unet("fp16", True)
    out = run_unet(unet, g)
    s2 = np.concatenate([g["sample"][:1], g["sample"][:1], g["sample"][1:], g["sample"][1:]])
    e2 = np.concatenate([g["ehs"][:1], g["ehs"][:1], g["ehs"][1:], g["ehs"][1:]])
    out2 = unet(torch.from_numpy(s2), float(g["timestep"]), torch.from_numpy(e2), torch.from_numpy(np.concatenate([g["added_time_ids"]] * 2)),
                torch.from_numpy(np.concatenate([g["action_ids"]] * 2))).cpu().numpy()
    assert np.array_equal(out2[0], out[0]) and np.array_equal(out2[1], out[0])
    assert np.array_equal(out2[2], out[1]) and np.array_equal(out2[3], out[1])
    p = golden("pipeline_full_16x32.npz")
    args = (torch.from_numpy(p["image_latents"]), torch.from_numpy(p["image_embeddings"]), torch.from_numpy(p["latent_noise"]), p["actions"])
    eager = SVDDenoiser(unet, use_graph=False).denoise(*args, num_steps=3).cpu()
    dg = SVDDenoiser(unet, use_graph=True)
    replay = dg.denoise(*args, num_steps=3).cpu()
    assert dg.graph_error is None and torch.equal(eager, replay)


# relative latent error (rms) of the HIP loop after the reference's full schedule, per build: (vs the reference's fp32
# trajectory, vs the reference on the same rounded weights).  1.25 x the measurement of round 4
# (profiles/r08b_res32_tests.log: 5.08e-4 / 4.47e-4, 6.61e-4 / 6.26e-4, 4.11e-3 / 3.55e-3, 5.25e-3 / 4.92e-3); both fp16
# configurations are inside north_star's 1e-3 relative latent error on the reference's own trajectory.
LOOP_GATES = {
    ("fp16", True): (6.4e-4, 5.6e-4),
    ("fp16", False): (8.3e-4, 7.9e-4),
    ("bf16", True): (5.2e-3, 4.5e-3),
    ("bf16", False): (6.6e-3, 6.2e-3),
}


def _loop(name, res32, p, steps, unet=None):
    from wiw_amd.pipeline import SVDDenoiser

    unet = unet or full_unet(name, res32)
    traj = {}
    keep = set(int(k) for k in p["trajectory_steps"])
    lat = SVDDenoiser(unet, use_graph=False).denoise(
        torch.from_numpy(p["image_latents"]), torch.from_numpy(p["image_embeddings"]), torch.from_numpy(p["latent_noise"]),
        p["actions"], num_steps=steps, callback=lambda i, x: traj.__setitem__(i + 1, x.cpu().numpy().copy()) if (i + 1) in keep else None)
    return lat.cpu().numpy(), traj


@pytest.mark.parametrize("name,res32", list(LOOP_GATES))
def test_reference_trajectory_25_steps_served_width(name, res32, golden):
    """The HIP loop against `StableVideoDiffusionPipeline.__call__` itself: 25 Euler steps, served-width UNet, 16x32 latent,
    T = 14, CFG — the fixture holds the reference's latents (fp32, and fp32 arithmetic on 16-bit-rounded weights)."""
    p = golden("pipeline_full_16x32.npz")
    steps = int(p["num_steps"])
    lat, traj = _loop(name, res32, p, steps)
    mx, rms = rel(lat, p["latents_out"])
    mx_w, rms_w = rel(lat, p[f"latents_out_{name}_weights"])
    floor = rel(p[f"latents_out_{name}_weights"], p["latents_out"])[1]
    per = " ".join(f"{k}:{rel(traj[k], p['trajectory'][i][None])[1]:.2e}" for i, k in enumerate(int(k) for k in p["trajectory_steps"]))
    print(f"[tolerance] 25-step served-width rollout vs the reference pipeline, {name}{label32(res32)}: "
          f"relative latent error rms={rms:.3e} max={mx:.3e} | vs the same-weights reference rms={rms_w:.3e} | weight-rounding floor "
          f"{floor:.3e} | per step {per}")
    assert np.isfinite(lat).all()
    gate, gate_w = LOOP_GATES[(name, res32)]
    assert rms <= gate and rms_w <= gate_w


# 1.25 x the round-4 measurement (profiles/r10y_gpu_suite_*.log: 6.42e-4, 9.04e-4, 7.69e-3); both fp16 configurations are
# inside north_star's 1e-3 on this workload too
CONFIG0_GATES = {("fp16", True): 8.0e-4, ("fp16", False): 1.0e-3, ("bf16", False): 9.6e-3}


@pytest.mark.parametrize("name,res32", list(CONFIG0_GATES))
def test_baseline_config0_workload(name, res32, golden):
    """BASELINE.json configs[0] on the HIP path: the 256x256x8-frame, 10-step rollout of FTsvd/eval_inference.py (latent 32x32,
    served-width UNet built for 8 frames) against the reference pipeline's own latents; twice, bit-identical."""
    import wiw_amd  # noqa: F401
    from wiw_amd.config import UNetConfig
    from wiw_amd.unet import UNetHIP
    from wiw_amd.weights import random_state_dict

    p = golden("pipeline_config0_32x32.npz")
    T = int(p["num_frames"])
    assert T == 8 and p["latent_noise"].shape == (1, 8, 4, 32, 32) and int(p["num_steps"]) == 10
    cfg = UNetConfig(num_frames=T, action_input_channel=T)
    sd = {k: torch.from_numpy(v) for k, v in random_state_dict(cfg, int(p["weight_seed"])).items()}
    unet = UNetHIP(cfg, sd, DEV, hip=get_hip(name), residual_fp32=res32)
    lat, _ = _loop(name, res32, p, 10, unet=unet)
    lat2, _ = _loop(name, res32, p, 10, unet=unet)
    mx, rms = rel(lat, p["latents_out"])
    print(f"[tolerance] BASELINE config 0 (256x256x8, 10 steps) vs the reference pipeline, {name}{label32(res32)}: "
          f"relative latent error rms={rms:.3e} max={mx:.3e}")
    assert np.isfinite(lat).all() and np.array_equal(lat, lat2)
    assert rms <= CONFIG0_GATES[(name, res32)]
    del unet
    torch.cuda.empty_cache()


# ----------------------------------------------------------------------------------------------
# the BENCHMARKED size: one forward of the reference UNet at sample (2, 14, 8, 72, 128) = 576x1024x14 with CFG
# (tests/golden/unet_northstar_72x128.npz, `oracle/make_golden.py unet_northstar`; reference:
# dp/models/unets/unet_spatio_temporal_condition.py:402-575).  S = 9216 keys in the spatial softmax, 18 432-site temporal
# batches, GroupNorm units of 2.6 M elements: what the 16x32 / 32x32 fixtures do not exercise (VERDICT r4 "missing" 1).
# ----------------------------------------------------------------------------------------------
def northstar_inputs(g):
    """The fixture stores no inputs (8 MB): `unet_inputs(cfg, 1, 72, 128, seed)` of oracle/make_golden.py, restated; numpy's
    legacy RandomState is frozen, the float64 checksums pin the draw."""
    h, w = (int(v) for v in g["latent_hw"])
    rs = np.random.RandomState(int(g["input_seed"]))
    sample = rs.standard_normal((2, 14, 8, h, w)).astype(np.float32)
    ehs = rs.standard_normal((2, 1, 1024)).astype(np.float32)
    ehs[:1] = 0
    sample[:1, :, 4:] = 0
    assert abs(float(sample.astype(np.float64).sum()) - float(g["sample_checksum"])) < 1e-6
    assert abs(float(np.abs(sample.astype(np.float64)).sum()) - float(g["sample_abs_checksum"])) < 1e-6
    return sample, ehs


# rms gates at the benchmarked size: (vs the reference's fp32 output, vs the reference in fp32 math on the same rounded weights).
# This is ONE v-prediction at sigma = 15.6, not a latent: north_star's "1e-3 relative latent error" is the trajectory test below.
# fp16 + FULL fp32 stream: 1e-3 all the same (unchanged since round 5); fp16 + block-level fp32 stream (round 6): 1.2 x measured;
# the others 1.2 x the round-5 measurement (profiles/r12h_gpu_partial_with_northstar_parity.log, r20c_res32_rb_tests.log)
NORTHSTAR_GATES = {
    ("fp16", "full"): (1.0e-3, 1.0e-3),    # measured 9.12e-4 / 9.15e-4 (16x32: 8.61e-4; weight-rounding floor 7.8e-4)
    ("fp16", True): (1.2e-3, 1.22e-3),     # measured 1.003e-3 / 1.019e-3 (16x32: 9.47e-4)
    ("fp16", False): (1.55e-3, 1.57e-3),   # measured 1.283e-3 / 1.300e-3
    ("bf16", False): (1.26e-2, 1.25e-2),   # measured 1.044e-2 / 1.041e-2 (weight-rounding floor 7.1e-3)
}


@pytest.mark.parametrize("name,res32", list(NORTHSTAR_GATES))
def test_unet_north_star_size_against_the_reference(name, res32, golden):
    g = golden("unet_northstar_72x128.npz")
    assert int(g["weight_seed"]) == 4
    sample, ehs = northstar_inputs(g)
    unet = full_unet(name, res32)
    out = unet(torch.from_numpy(sample), float(g["timestep"]), torch.from_numpy(ehs), torch.from_numpy(g["added_time_ids"]),
               torch.from_numpy(g["action_ids"])).cpu().numpy()
    ref = g["out"]
    ref_w = ref + g[f"diff_{name}_weights_fp32_math"].astype(np.float32)
    mx, rms = rel(out, ref)
    mx_w, rms_w = rel(out, ref_w)
    floor = rel(ref_w, ref)[1]
    g16 = golden("unet_full_16x32.npz")
    rms16 = rel(run_unet(unet, g16), g16["out"])[1]
    print(f"[tolerance] unet at the BENCHMARKED size 72x128x14 (S = 9216), {name}{label32(res32)}: "
          f"vs reference fp32 rms={rms:.3e} max={mx:.3e} | vs reference on the SAME {name}-rounded weights (fp32 math) rms={rms_w:.3e} "
          f"max={mx_w:.3e} | weight-rounding floor {floor:.3e} | the same build at 16x32: {rms16:.3e}")
    assert np.isfinite(out).all()
    gate, gate_w = NORTHSTAR_GATES[(name, res32)]
    assert rms <= gate, f"{name} res32={res32}: rms {rms:.3e} vs the reference's fp32 output at 72x128 exceeds {gate:.2e}"
    assert rms_w <= gate_w, f"{name} res32={res32}: rms {rms_w:.3e} vs the same-weights reference at 72x128 exceeds {gate_w:.2e}"
    torch.cuda.empty_cache()


# ----------------------------------------------------------------------------------------------
# north_star's parity sentence on BASELINE config 1 itself: the reference pipeline's OWN 25-step trajectory at the benchmarked
# size (tests/golden/pipeline_northstar_72x128.npz, `oracle/make_golden.py pipeline_northstar`: StableVideoDiffusionPipeline
# .__call__, pipeline_stable_video_diffusion.py:383-638, output_type='latent', 576x1024x14, B = 1 with CFG, fp32; 25 reference
# forwards = 2 h of the build container's cores).  VERDICT r5 item 2: the 16x32 trajectory says the error is made in steps
# 15-25 (sigma < 3) — not at the sigma the one-forward 72x128 fixture samples.
# ----------------------------------------------------------------------------------------------
def northstar_pipeline_noise(p):
    """The fixture stores no noise (2 MB): the generator's draws restated in its order (image, image noise, latent noise) on
    numpy's frozen legacy RandomState; float64 checksums pin the draw."""
    h, w = (int(v) for v in p["latent_hw"])
    T = int(p["num_frames"])
    rs = np.random.RandomState(int(p["input_seed"]))
    rs.randint(0, 256, size=(8 * h, 8 * w, 3), dtype=np.uint8)
    rs.standard_normal((1, 3, 8 * h, 8 * w))
    noise = rs.standard_normal((1, T, 4, h, w)).astype(np.float32)
    assert abs(float(noise.astype(np.float64).sum()) - float(p["noise_checksum"])) < 1e-6
    assert abs(float(np.abs(noise.astype(np.float64)).sum()) - float(p["noise_abs_checksum"])) < 1e-6
    return noise


# (rms, max) gates of the relative latent error after 25 steps at 72x128 — north_star's sentence on BASELINE config 1.  fp16 + fp32
# stream, block-level (the mode) and full: 1e-3 on BOTH norms (measured 5.47e-4 / 8.58e-4 and 5.12e-4 / 8.56e-4); fp16: 1e-3 on the
# rms (6.75e-4), 1.2 x measured on the max norm (1.09e-3: the one norm of the 16-bit configurations that sits above 1e-3 at this
# size — DESIGN 5); bf16: 1.2 x measured (5.21e-3 / 7.65e-3).  Measurements: profiles/r19i_northstar_trajectory.log, r20b_*
NORTHSTAR_LOOP_GATES = {
    ("fp16", True): (1.0e-3, 1.0e-3),
    ("fp16", "full"): (1.0e-3, 1.0e-3),
    ("fp16", False): (1.0e-3, 1.32e-3),
    ("bf16", False): (6.3e-3, 9.2e-3),
}


@pytest.mark.parametrize("name,res32", list(NORTHSTAR_LOOP_GATES))
def test_reference_trajectory_25_steps_at_the_benchmarked_size(name, res32, golden):
    from wiw_amd.pipeline import SVDDenoiser

    if not os.path.exists(os.path.join(os.path.dirname(__file__), "golden", "pipeline_northstar_72x128.npz")):
        pytest.skip("tests/golden/pipeline_northstar_72x128.npz not generated yet (oracle/make_golden.py pipeline_northstar, ~2 h)")
    p = golden("pipeline_northstar_72x128.npz")
    assert int(p["weight_seed"]) == 4 and int(p["num_steps"]) == 25 and tuple(int(v) for v in p["latent_hw"]) == (72, 128)
    noise = northstar_pipeline_noise(p)
    keep = [int(k) for k in p["trajectory_steps"]]
    traj = {}
    unet = full_unet(name, res32)
    lat = SVDDenoiser(unet, use_graph=False).denoise(
        torch.from_numpy(p["image_latents"]), torch.from_numpy(p["image_embeddings"]), torch.from_numpy(noise), p["actions"],
        num_steps=25, callback=lambda i, x: traj.__setitem__(i + 1, x.cpu().numpy().copy()) if (i + 1) in keep else None).cpu().numpy()
    mx, rms = rel(lat, p["latents_out"])
    per = " ".join(f"{k}:{rel(traj[k], p['trajectory'][i][None])[1]:.2e}/{rel(traj[k], p['trajectory'][i][None])[0]:.2e}" for i, k in enumerate(keep))
    print(f"[tolerance] 25-step rollout at the BENCHMARKED size 576x1024x14 vs the reference pipeline's own latents, {name}"
          f"{label32(res32)}: relative latent error rms={rms:.3e} max={mx:.3e} | per step rms/max {per}")
    assert np.isfinite(lat).all()
    g_rms, g_max = NORTHSTAR_LOOP_GATES[(name, res32)]
    assert rms <= g_rms, f"{name} res32={res32}: rms {rms:.3e} > {g_rms:.1e}"
    assert mx <= g_max, f"{name} res32={res32}: max {mx:.3e} > {g_max:.1e}"
    torch.cuda.empty_cache()
