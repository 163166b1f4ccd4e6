"""Parity of every HIP kernel (through the C ABI) against the CPU oracle on seeded inputs.  GPU only.

Tolerances (stated here, used below):
  * kernels compute with bf16 operands and fp32 accumulation and store bf16; against the oracle run in
    fp32 on the SAME bf16-rounded inputs the only differences are fp32 summation order and the final
    bf16 rounding (2^-9 relative).  Gate: max|err| <= 1.2e-2 * max|ref| and rms(err) <= 4e-3 * rms(ref).
  * fp32-output kernels (Euler step, GroupNorm scale/shift): 1e-5 relative.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import svd_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def hip():
    import wiw_amd  # noqa: F401
    from wiw_amd.hip import Hip

    return Hip(torch.device(DEV))


def bf(x):  # round to bf16 and back (what the kernels see)
    return x.to(torch.bfloat16).float()


def dev_bf(x):
    return x.to(DEV, torch.bfloat16).contiguous()


def dev_f(x):
    return x.to(DEV, torch.float32).contiguous()


def check(out, ref, max_tol=1.2e-2, rms_tol=4e-3, what=""):
    out = out.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert out.shape == ref.shape, (what, out.shape, ref.shape)
    assert torch.isfinite(out).all(), f"{what}: non-finite output"
    err = (out - ref).abs()
    mx = float(err.max() / (ref.abs().max() + 1e-30))
    rms = float(err.pow(2).mean().sqrt() / (ref.pow(2).mean().sqrt() + 1e-30))
    print(f"[parity] {what}: max_rel={mx:.3e} rms_rel={rms:.3e}")
    assert mx <= max_tol and rms <= rms_tol, f"{what}: max_rel={mx:.3e} rms_rel={rms:.3e}"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# ----------------------------------------------------------------------------------------------
# GEMM, dense mode
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(128, 160, 64), (300, 320, 320), (28, 1280, 320), (4032, 1280, 1280),
                                   (1000, 200, 128), (128, 200, 128), (130, 8, 64), (64, 4, 192)])
def test_gemm_dense_plain(hip, M, N, K):
    from wiw_amd import hip as H

    a, w = bf(rnd(M, K, seed=1)), bf(rnd(N, K, seed=2) / math.sqrt(K))
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    hip.gemm(dev_bf(a), dev_bf(w), out, M=M, N=N, K=K, C1=K)
    check(out, a @ w.t(), what=f"gemm {M}x{N}x{K}")
    # asymmetric A = I check (catches a transposed C write)
    if M == K:
        eye = torch.eye(M)
        hip.gemm(dev_bf(eye), dev_bf(w), out, M=M, N=N, K=K, C1=K)
        check(out, w.t(), what="gemm identity")
    out32 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    hip.gemm(dev_bf(a), dev_bf(w), out32, M=M, N=N, K=K, C1=K, epilogue=H.EPI_OUT_F32)
    check(out32, a @ w.t(), max_tol=2e-5, rms_tol=2e-6, what="gemm f32 out")


def test_gemm_epilogues(hip):
    from wiw_amd import hip as H

    M, N, K, rpv = 700, 320, 256, 50
    a, w = bf(rnd(M, K, seed=1)), bf(rnd(N, K, seed=2) / 16)
    bias, rowvec = rnd(N, seed=3), rnd(M // rpv, N + 24, seed=4)
    r1, r2 = bf(rnd(M, N, seed=5)), bf(rnd(M, N, seed=6))
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    rv_dev = dev_f(rowvec)
    hip.gemm(dev_bf(a), dev_bf(w), out, M=M, N=N, K=K, C1=K, bias=dev_f(bias), rowvec=rv_dev[:, 8:], rowvec_ld=N + 24,
             rows_per_vec=rpv, res1=dev_bf(r1), ldr1=N, beta1=0.7, res2=dev_bf(r2), ldr2=N, beta2=0.3, alpha=0.6)
    ref = 0.6 * (a @ w.t() + bias + rowvec[:, 8:8 + N].repeat_interleave(rpv, 0)) + 0.7 * r1 + 0.3 * r2
    check(out, ref, what="gemm epilogue mix")
    hip.gemm(dev_bf(a), dev_bf(w), out, M=M, N=N, K=K, C1=K, bias=dev_f(bias), epilogue=H.EPI_SILU)
    check(out, F.silu(a @ w.t() + bias), what="gemm silu")
    # in-place residual (out aliases res1), as used by the attention out-projection
    res = dev_bf(r1).clone()
    hip.gemm(dev_bf(a), dev_bf(w), res, M=M, N=N, K=K, C1=K, bias=dev_f(bias), res1=res, ldr1=N, beta1=1.0)
    check(res, a @ w.t() + bias + r1, what="gemm in-place residual")


@pytest.mark.parametrize("Cn", [64, 320])
def test_gemm_geglu(hip, Cn):
    from wiw_amd import hip as H
    from wiw_amd.unet import pack_geglu

    M = 520
    a = bf(rnd(M, Cn, seed=1))
    w, b = rnd(8 * Cn, Cn, seed=2) / math.sqrt(Cn), rnd(8 * Cn, seed=3)
    wp, bp, n_half = pack_geglu(w, b)
    out = torch.empty(M, 4 * Cn, dtype=torch.bfloat16, device=DEV)
    hip.gemm(dev_bf(a), dev_bf(wp), out, M=M, N=wp.shape[0], K=Cn, C1=Cn, bias=dev_f(bp), epilogue=H.EPI_GEGLU, n_out=4 * Cn)
    h = a @ bf(w).t() + b
    val, gate = h.chunk(2, dim=-1)
    check(out, val * F.gelu(gate), what=f"gemm geglu C={Cn}")


def test_gemm_concat_and_swapped(hip):
    M, C1, C2, N = 390, 128, 64, 192
    a1, a2, w = bf(rnd(M, C1, seed=1)), bf(rnd(M, C2, seed=2)), bf(rnd(N, C1 + C2, seed=3) / 14)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    hip.gemm(dev_bf(a1), dev_bf(w), out, M=M, N=N, K=C1 + C2, C1=C1, A2=dev_bf(a2), C2=C2)
    check(out, torch.cat([a1, a2], 1) @ w.t(), what="gemm concat-K")
    # swapped operands -> V^T layout used by spatial attention
    Cn, Mt = 128, 1000
    x, wv = bf(rnd(Mt, Cn, seed=4)), bf(rnd(Cn, Cn, seed=5) / 11)
    vt = torch.empty(Cn, Mt, dtype=torch.bfloat16, device=DEV)
    hip.gemm(dev_bf(wv), dev_bf(x), vt, M=Cn, N=Mt, K=Cn, C1=Cn)
    check(vt, (x @ wv.t()).t(), what="gemm swapped (V^T)")


# ----------------------------------------------------------------------------------------------
# implicit-GEMM convolutions
# ----------------------------------------------------------------------------------------------
def nhwc(x):  # (N,C,H,W) -> token-major [N*H*W, C]
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()


def from_nhwc(t, n, h, w):
    return t.float().cpu().reshape(n, h, w, -1).permute(0, 3, 1, 2)


@pytest.mark.parametrize("n,cin,cout,h,w", [(3, 64, 96, 8, 16), (2, 128, 320, 9, 16), (5, 64, 64, 4, 8), (1, 64, 4, 8, 8)])
def test_conv3x3(hip, n, cin, cout, h, w):
    from wiw_amd import hip as H

    x = bf(rnd(n, cin, h, w, seed=1))
    wt = bf(rnd(cout, cin, 3, 3, seed=2) / math.sqrt(9 * cin))
    b = rnd(cout, seed=3)
    wk = dev_bf(wt.permute(0, 2, 3, 1).reshape(cout, -1))
    M = n * h * w
    out = torch.empty(M, cout, dtype=torch.bfloat16, device=DEV)
    hip.gemm(dev_bf(nhwc(x)), wk, out, M=M, N=cout, K=9 * cin, C1=cin, mode=H.A_CONV3X3, H=h, Wd=w, bias=dev_f(b))
    check(from_nhwc(out, n, h, w), F.conv2d(x, wt, b, padding=1), what=f"conv3x3 {n}x{cin}->{cout}@{h}x{w}")


def test_conv_stride2_and_upsample(hip):
    from wiw_amd import hip as H

    n, c, h, w = 3, 64, 8, 16
    x = bf(rnd(n, c, h, w, seed=1))
    wt = bf(rnd(c, c, 3, 3, seed=2) / math.sqrt(9 * c))
    b = rnd(c, seed=3)
    wk = dev_bf(wt.permute(0, 2, 3, 1).reshape(c, -1))
    Mo = n * (h // 2) * (w // 2)
    out = torch.empty(Mo, c, dtype=torch.bfloat16, device=DEV)
    hip.gemm(dev_bf(nhwc(x)), wk, out, M=Mo, N=c, K=9 * c, C1=c, mode=H.A_CONV3X3_S2, H=h // 2, Wd=w // 2, bias=dev_f(b))
    check(from_nhwc(out, n, h // 2, w // 2), F.conv2d(x, wt, b, stride=2, padding=1), what="conv3x3 stride 2")
    Mo = n * 4 * h * w
    out = torch.empty(Mo, c, dtype=torch.bfloat16, device=DEV)
    hip.gemm(dev_bf(nhwc(x)), wk, out, M=Mo, N=c, K=9 * c, C1=c, mode=H.A_CONV3X3_UP, H=2 * h, Wd=2 * w, bias=dev_f(b))
    ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), wt, b, padding=1)
    check(from_nhwc(out, n, 2 * h, 2 * w), ref, what="nearest-up + conv3x3")


@pytest.mark.parametrize("B,T,c,h,w", [(2, 4, 64, 4, 8), (1, 14, 128, 3, 8)])
def test_conv_temporal(hip, B, T, c, h, w):
    from wiw_amd import hip as H

    x = bf(rnd(B * T, c, h, w, seed=1))
    wt = bf(rnd(c, c, 3, 1, 1, seed=2) / math.sqrt(3 * c))
    b = rnd(c, seed=3)
    wk = dev_bf(wt[:, :, :, 0, 0].permute(0, 2, 1).reshape(c, -1))
    M = B * T * h * w
    out = torch.empty(M, c, dtype=torch.bfloat16, device=DEV)
    hip.gemm(dev_bf(nhwc(x)), wk, out, M=M, N=c, K=3 * c, C1=c, mode=H.A_CONV_T3, H=h, Wd=w, T=T, bias=dev_f(b))
    x5 = x.reshape(B, T, c, h, w).permute(0, 2, 1, 3, 4)
    ref = F.conv3d(x5, wt, b, padding=(1, 0, 0)).permute(0, 2, 1, 3, 4).reshape(B * T, c, h, w)
    check(from_nhwc(out, B * T, h, w), ref, what=f"temporal conv T={T}")


# ----------------------------------------------------------------------------------------------
# attention
# ----------------------------------------------------------------------------------------------
def sdpa(q, k, v, heads):
    Bn, S, C = q.shape
    d = C // heads
    q, k, v = (t.reshape(Bn, -1, heads, d).transpose(1, 2) for t in (q, k, v))
    w = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(d), dim=-1)
    return (w @ v).transpose(1, 2).reshape(Bn, S, C)


@pytest.mark.parametrize("frames,S,heads", [(2, 128, 1), (3, 144, 2), (1, 8, 2), (2, 200, 5), (1, 1024, 1)])
def test_attn_spatial(hip, frames, S, heads):
    C = heads * 64
    q, k, v = (bf(rnd(frames, S, C, seed=s, scale=sc)) for s, sc in ((1, 1.5), (2, 1.5), (3, 1.0)))
    qk = dev_bf(torch.cat([q, k], -1).reshape(frames * S, 2 * C))
    vt = dev_bf(v.reshape(frames * S, C).t())
    o = torch.empty(frames * S, C, dtype=torch.bfloat16, device=DEV)
    hip.attn_spatial(qk, 2 * C, C, vt, frames * S, o, C, frames, S, heads, 0.125)
    check(o.reshape(frames, S, C), sdpa(q, k, v, heads), max_tol=2e-2, rms_tol=8e-3, what=f"attn_spatial S={S} h={heads}")


PRESCALE = math.log2(math.e) / 8.0   # what unet.py folds into the to_q rows (ATTN_PRESCALE)


@pytest.mark.parametrize("frames,S,heads", [(2, 256, 1), (1, 1024, 3), (3, 384, 2), (2, 144, 2), (1, 576, 5), (1, 8, 2), (2, 200, 1)])
def test_attn_spatial_prescaled(hip, frames, S, heads):
    """wiw_attn_spatial_ps_bf16: the 32x32x16 kernel (S % 128 == 0, S >= 256) and its fallback, against fp32 softmax attention
    on the SAME rounded operands (the pre-scaled Q is what the kernel is handed; the reference divides the factor out)."""
    C = heads * 64
    q, k, v = (rnd(frames, S, C, seed=s, scale=sc) for s, sc in ((1, 1.5), (2, 1.5), (3, 1.0)))
    qs, k, v = bf(q * PRESCALE), bf(k), bf(v)
    qk = dev_bf(torch.cat([qs, k], -1).reshape(frames * S, 2 * C))
    vt = dev_bf(v.reshape(frames * S, C).t())
    o = torch.empty(frames * S, C, dtype=torch.bfloat16, device=DEV)
    hip.attn_spatial_ps(qk, 2 * C, C, vt, frames * S, o, C, frames, S, heads)
    check(o.reshape(frames, S, C), sdpa(qs / PRESCALE, k, v, heads), max_tol=2e-2, rms_tol=8e-3, what=f"attn_spatial_ps S={S} h={heads}")


@pytest.mark.parametrize("S", [256, 1024])
def test_attn_spatial_prescaled_raises_the_reference(hip, S):
    """Keys with huge scores LATE in the sequence (and one in the second tile) force the 32x32x16 kernel's rare branch: the
    reference is raised on the integer grid, O / l / the pending P operands / the block in flight are rescaled (T13)."""
    frames, heads, C = 2, 1, 64
    q, k, v = rnd(frames, S, C, seed=1), rnd(frames, S, C, seed=2), rnd(frames, S, C, seed=3)
    k[0, S - 56] = q[0, 17] * 6.0
    k[0, 70] = q[0, 17] * 3.0
    k[1, S // 2 + 5] = q[1, 100] * 9.0
    k[1, S // 2 + 37] = q[1, 101] * 12.0
    qs, k, v = bf(q * PRESCALE), bf(k), bf(v)
    qk = dev_bf(torch.cat([qs, k], -1).reshape(frames * S, 2 * C))
    vt = dev_bf(v.reshape(frames * S, C).t())
    o = torch.empty(frames * S, C, dtype=torch.bfloat16, device=DEV)
    hip.attn_spatial_ps(qk, 2 * C, C, vt, frames * S, o, C, frames, S, heads)
    assert torch.isfinite(o.float()).all()
    check(o.reshape(frames, S, C), sdpa(qs / PRESCALE, k, v, heads), max_tol=2e-2, rms_tol=8e-3, what="attn_spatial_ps rescale")


def test_attn_spatial_prescaled_long_sequence_many_blocks(hip):
    """A level-1-sized problem on the 32x32x16 kernel: 36 KV tiles per block, 1152 blocks over the XCD remap; and bit-exact
    repeatability (no atomics, fixed order)."""
    frames, S, heads = 8, 2304, 8
    C = heads * 64
    qkv = rnd(frames * S, 3 * C, seed=21)
    qs, k, v = bf(qkv[:, :C] * PRESCALE), bf(qkv[:, C:2 * C]), bf(qkv[:, 2 * C:])
    qk = dev_bf(torch.cat([qs, k], dim=1))
    vt = dev_bf(v.t().contiguous())
    o = torch.empty(frames * S, C, dtype=torch.bfloat16, device=DEV)
    o2 = torch.empty_like(o)
    hip.attn_spatial_ps(qk, 2 * C, C, vt, frames * S, o, C, frames, S, heads)
    hip.attn_spatial_ps(qk, 2 * C, C, vt, frames * S, o2, C, frames, S, heads)
    ref = sdpa((qs / PRESCALE).reshape(frames, S, C), k.reshape(frames, S, C), v.reshape(frames, S, C), heads).reshape(frames * S, C)
    check(o, ref, what="attn_spatial_ps 8 x 2304 x 8 heads")
    assert torch.equal(o, o2)


def test_attn_spatial_online_softmax_rescale(hip):
    """One key with a huge score in a LATE tile forces the running-max rescale branch (every tile)."""
    frames, S, heads, C = 1, 256, 1, 64
    q, k, v = bf(rnd(frames, S, C, seed=1)), bf(rnd(frames, S, C, seed=2)), bf(rnd(frames, S, C, seed=3))
    k[0, 200] = q[0, 17] * 6.0
    k[0, 70] = q[0, 17] * 3.0
    qk = dev_bf(torch.cat([q, k], -1).reshape(S, 2 * C))
    vt = dev_bf(v.reshape(S, C).t())
    o = torch.empty(S, C, dtype=torch.bfloat16, device=DEV)
    hip.attn_spatial(qk, 2 * C, C, vt, S, o, C, frames, S, heads, 0.125)
    check(o.reshape(frames, S, C), sdpa(q, k, v, heads), max_tol=2e-2, rms_tol=8e-3, what="attn_spatial rescale")


@pytest.mark.parametrize("B,T,S,heads", [(2, 4, 32, 1), (1, 14, 72, 5), (3, 16, 8, 2), (2, 1, 8, 1)])
def test_attn_temporal(hip, B, T, S, heads):
    C = heads * 64
    q, k, v = (bf(rnd(B, T, S, C, seed=s, scale=1.5)) for s in (1, 2, 3))
    qkv = dev_bf(torch.cat([q, k, v], -1).reshape(B * T * S, 3 * C))
    o = torch.empty(B * T * S, C, dtype=torch.bfloat16, device=DEV)
    hip.attn_temporal(qkv, 3 * C, o, C, B, T, S, heads, 0.125)

    def perm(t):  # (B,T,S,C) -> (B*S, T, C)
        return t.permute(0, 2, 1, 3).reshape(B * S, T, C)

    ref = sdpa(perm(q), perm(k), perm(v), heads).reshape(B, S, T, C).permute(0, 2, 1, 3)
    check(o.reshape(B, T, S, C), ref, max_tol=2e-2, rms_tol=8e-3, what=f"attn_temporal T={T} S={S} h={heads}")


@pytest.mark.parametrize("heads,P,items,rpi", [(5, 3, 3, 1000), (5, 1, 2, 257), (10, 8, 2, 2304), (20, 2, 4, 144), (5, 5, 2, 14 * 576)])
def test_cross_attention_over_a_few_keys(hip, heads, P, items, rpi):
    """`wiw_cross_attn_fewkeys_bf16` (ABI 16, csrc/cross_attn.hip): softmax(Q K^T / 8) V with Sk = P conditioning tokens per
    item, the rows of an item sharing its keys — against torch's fp32 attention on the same 16-bit operands; strided Q / O;
    bit-repeatable; P outside 1..8 refused."""
    C = heads * 64
    rows = items * rpi
    q = bf(rnd(rows, C + 64, seed=1, scale=1.5))
    k, v = bf(rnd(items, P, C, seed=2, scale=1.2)), bf(rnd(items, P, C, seed=3))
    o = torch.full((rows, C + 8), float("nan"), dtype=torch.bfloat16, device=DEV)
    qd = dev_bf(q)
    hip.cross_attn_fewkeys(qd, C + 64, dev_bf(k), dev_bf(v), o, C + 8, rows, rpi, heads, P, 0.125)
    qh = q[:, :C].reshape(items, rpi, heads, 64).permute(0, 2, 1, 3)
    kh, vh = (t.reshape(items, P, heads, 64).permute(0, 2, 1, 3) for t in (k, v))
    w = torch.softmax(qh @ kh.transpose(-1, -2) * 0.125, dim=-1)
    ref = (w @ vh).permute(0, 2, 1, 3).reshape(rows, C)
    check(o[:, :C], ref, max_tol=1.2e-2, rms_tol=3e-3, what=f"cross attention heads={heads} P={P} items={items} rows/item={rpi}")
    assert torch.isnan(o[:, C:].float()).all(), "columns beyond heads * 64 must not be written"
    o2 = torch.empty_like(o)
    hip.cross_attn_fewkeys(qd, C + 64, dev_bf(k), dev_bf(v), o2, C + 8, rows, rpi, heads, P, 0.125)
    assert torch.equal(o[:, :C], o2[:, :C])
    s = torch.cuda.current_stream().cuda_stream
    assert hip.lib.wiw_cross_attn_fewkeys_bf16(s, qd.data_ptr(), C + 64, qd.data_ptr(), qd.data_ptr(), o.data_ptr(), C + 8, rows, rpi,
                                               heads, 9, 0.125) != 0 and b"P <= 8" in hip.lib.wiw_last_error()


# ----------------------------------------------------------------------------------------------
# normalisation
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,c1,c2,h,w,unit_frames", [(4, 64, 0, 4, 8, 1), (4, 320, 0, 6, 8, 1), (4, 128, 64, 4, 8, 1),
                                                     (4, 1280, 640, 2, 4, 1), (4, 64, 0, 4, 8, 2), (6, 2560, 0, 2, 2, 3)])
def test_groupnorm(hip, n, c1, c2, h, w, unit_frames):
    C = c1 + c2
    x = bf(rnd(n, C, h, w, seed=1) * 2 + 0.5)
    gamma, beta = 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
    t = nhwc(x)
    x1 = dev_bf(t[:, :c1])
    x2 = dev_bf(t[:, c1:]) if c2 else None
    rows = n * h * w
    for silu in (True, False):
        out = hip.groupnorm(x1, c1, x2, c2, rows, unit_frames * h * w, dev_f(gamma), dev_f(beta), 1e-5, silu)
        if unit_frames == 1:
            ref = F.group_norm(x, 32, gamma, beta, 1e-5)
        else:  # 5-D GroupNorm of TemporalResnetBlock: statistics over the frames of a batch item too
            x5 = x.reshape(n // unit_frames, unit_frames, C, h, w).permute(0, 2, 1, 3, 4)
            ref = F.group_norm(x5, 32, gamma, beta, 1e-5).permute(0, 2, 1, 3, 4).reshape(n, C, h, w)
        ref = F.silu(ref) if silu else ref
        check(from_nhwc(out, n, h, w), ref, what=f"groupnorm C={c1}+{c2} unit={unit_frames} silu={silu}")
        out4 = hip.groupnorm_unfused(x1, c1, x2, c2, rows, unit_frames * h * w, dev_f(gamma), dev_f(beta), 1e-5, silu)
        check(from_nhwc(out4, n, h, w), ref, what=f"groupnorm (4 entry points) C={c1}+{c2} silu={silu}")


@pytest.mark.parametrize("n,c1,c2,h,w", [(6, 1280, 0, 18, 32), (6, 1280, 0, 9, 16), (4, 1280, 1280, 9, 16), (3, 1280, 1280, 18, 32)])
def test_groupnorm_onepass(hip, n, c1, c2, h, w):
    """ABI 15: GroupNorm(32) (+SiLU) of the two inner levels in ONE pass (a frame's 160-channel slab in registers).  Against
    torch's fp32 group_norm, against the two-launch path (statistics + apply) on the same tensors, bit-identical over repeats;
    the dispatcher takes it exactly where wiw_groupnorm_onepass_ok says so."""
    C = c1 + c2
    x = bf(rnd(n, C, h, w, seed=11) * 2 + rnd(1, C, 1, 1, seed=12) * 3)        # per-channel offsets: |mean| >> std in some groups
    gamma, beta = 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
    t = nhwc(x)
    x1 = dev_bf(t[:, :c1])
    x2 = dev_bf(t[:, c1:]) if c2 else None
    rows = n * h * w
    assert hip.lib.wiw_groupnorm_onepass_ok(c1, c2, rows, h * w) == 1
    for silu in (True, False):
        ref = F.group_norm(x, 32, gamma, beta, 1e-5)
        ref = F.silu(ref) if silu else ref
        out = hip.groupnorm(x1, c1, x2, c2, rows, h * w, dev_f(gamma), dev_f(beta), 1e-5, silu)
        check(from_nhwc(out, n, h, w), ref, what=f"one-pass groupnorm C={c1}+{c2} {h}x{w} silu={silu}")
        again = hip.groupnorm(x1, c1, x2, c2, rows, h * w, dev_f(gamma), dev_f(beta), 1e-5, silu)
        assert torch.equal(out, again)
        hip.gn_two_kernels = True
        try:
            two = hip.groupnorm(x1, c1, x2, c2, rows, h * w, dev_f(gamma), dev_f(beta), 1e-5, silu)
        finally:
            hip.gn_two_kernels = False
        d = (out.float() - two.float()).abs().max().item()
        print(f"[gn one pass] C={c1}+{c2} {h}x{w} silu={silu}: max |one pass - two launches| = {d:.3e}")
        assert d <= 0.05          # one 16-bit ulp at the largest magnitudes: the statistics differ in the last fp32 bits only
    # shapes the kernel does not take: refused by the C entry point, routed to the two launches by the dispatcher
    for (a, b, r, u) in ((320, 0, 6 * 576, 576), (1280, 640, 6 * 576, 576), (1280, 0, 6 * 2304, 2304), (640, 0, 576, 576)):
        assert hip.lib.wiw_groupnorm_onepass_ok(a, b, r, u) == 0
    bad = hip.lib.wiw_groupnorm_onepass(torch.cuda.current_stream().cuda_stream, x1.data_ptr(), 320, None, 0, rows, h * w,
                                        dev_f(gamma).data_ptr(), dev_f(beta).data_ptr(), 1e-5, 0, x1.data_ptr())
    assert bad != 0 and b"one-pass" in hip.lib.wiw_last_error()


def test_groupnorm_is_deterministic(hip):
    """No atomics in the statistics: the same call gives the same BYTES every time (and so does the whole pipeline)."""
    n, C, h, w = 6, 320, 24, 32
    x1 = dev_bf(nhwc(bf(rnd(n, C, h, w, seed=4) * 3 + 0.7)))
    g, b = dev_f(1 + 0.1 * rnd(C, seed=5)), dev_f(0.1 * rnd(C, seed=6))
    outs = [hip.groupnorm(x1, C, None, 0, n * h * w, h * w * u, g, b, 1e-5, True) for u in (1, 1, 1, 3, 3)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]) and torch.equal(outs[3], outs[4])


@pytest.mark.parametrize("units,rpu,C,rpb", [(3, 16384 + 48, 64, 16), (2, 5000, 320, 16), (5, 640, 128, 16), (1, 37, 32, 16),
                                            (4, 9216, 320, 128)])
def test_groupnorm_statistics_in_one_launch(hip, units, rpu, C, rpb):
    """ABI 12: the second reduction stage runs inside the statistics launch — the block that finishes last for a part of 32
    blocks merges the part, the block that finds the unit's parts complete merges those (up to 33 parts here: the part loop runs
    9 times; partial last parts; a unit of ONE block).  Against fp64 torch statistics, bit-identical over repeated launches
    (whichever blocks finish last), counters back at zero after every launch."""
    rows = units * rpu
    x = bf(rnd(rows, C, seed=3) * 2.0 + rnd(1, C, seed=4) * 5.0)          # per-channel offsets: |mean| >> std in some groups
    xd = dev_bf(x)
    stats = torch.zeros(units * 64, dtype=torch.float32, device=DEV)
    scratch = torch.empty(int(hip.lib.wiw_groupnorm_scratch_floats(rows, rpu, rpb)), dtype=torch.float32, device=DEV)
    ncnt = int(hip.lib.wiw_groupnorm_counters(rows, rpu, rpb))
    splits = -(-rpu // rpb)
    assert ncnt == units * (-(-splits // 32) + 1)
    cnt = torch.zeros(ncnt, dtype=torch.int32, device=DEV)
    s = torch.cuda.current_stream().cuda_stream
    runs = []
    for _ in range(4):
        stats.fill_(float("nan"))
        assert hip.lib.wiw_groupnorm_stats(s, xd.data_ptr(), C, None, 0, rows, rpu, rpb, stats.data_ptr(), scratch.data_ptr(),
                                           cnt.data_ptr()) == 0
        torch.cuda.synchronize()
        assert int(cnt.abs().sum()) == 0, "counters must be left at zero"
        runs.append(stats.clone())
    assert all(torch.equal(runs[0], r) for r in runs[1:])
    st = runs[0].view(units, 32, 2).cpu().double()
    ref = x.double().view(units, rpu, 32, C // 32).transpose(1, 2).reshape(units, 32, -1)
    m, v = ref.mean(-1), ref.var(-1, unbiased=False)
    assert float((st[..., 0] - m).abs().max() / m.abs().max()) <= 2e-6
    assert float(((st[..., 1] - v).abs() / v).max()) <= 2e-5


@pytest.mark.parametrize("rows,C", [(100, 64), (777, 320), (64, 1280), (50, 2048)])
def test_layernorm(hip, rows, C):
    x = bf(rnd(rows, C, seed=1) * 1.7 + 0.3)
    gamma, beta = 1 + 0.1 * rnd(C, seed=2), 0.1 * rnd(C, seed=3)
    out = hip.layernorm(dev_bf(x), rows, C, dev_f(gamma), dev_f(beta))
    check(out, F.layer_norm(x, (C,), gamma, beta, 1e-5), what=f"layernorm C={C}")
    rpv = 10
    av = rnd((rows + rpv - 1) // rpv, C, seed=4)
    s = torch.empty(rows, C, dtype=torch.bfloat16, device=DEV)
    out = hip.layernorm(dev_bf(x), rows, C, dev_f(gamma), dev_f(beta), addvec=dev_f(av), addvec_ld=C, rows_per_vec=rpv, sum_out=s)
    xs = bf(x + av.repeat_interleave(rpv, 0)[:rows])
    check(s, xs, max_tol=1e-6, rms_tol=1e-6, what="layernorm sum_out")
    check(out, F.layer_norm(xs, (C,), gamma, beta, 1e-5), what=f"layernorm+add C={C}")


# ----------------------------------------------------------------------------------------------
# loop-side kernels
# ----------------------------------------------------------------------------------------------
def test_prep_and_euler(hip, golden):
    B, T, h, w = 2, 4, 4, 8
    lat = rnd(B, T, 4, h, w, seed=1) * 30
    img = rnd(B, 4, h, w, seed=2)
    sigma = 26.7572
    X = torch.empty(2 * B * T * h * w, 64, dtype=torch.bfloat16, device=DEV)
    hip.prep_unet_input(dev_f(lat), dev_f(img), B, T, h * w, sigma, 64, X)
    inp = torch.cat([lat, lat]) / (sigma ** 2 + 1) ** 0.5
    il = torch.cat([torch.zeros_like(img), img])[:, None].repeat(1, T, 1, 1, 1)
    ref = torch.cat([inp, il], 2).permute(0, 1, 3, 4, 2).reshape(-1, 8)
    Xc = X.float().cpu()
    check(Xc[:, :8], bf(ref), max_tol=8e-3, rms_tol=4e-3, what="prep_unet_input")
    assert float(Xc[:, 8:].abs().max()) == 0.0
    # CFG + Euler against the scheduler fixture produced by the reference (3 steps, same v for u and c halves)
    g = golden("scheduler_tables.npz")
    sig = O.karras_sigmas(10)
    x = torch.from_numpy(g["step_x"][0])  # (1,2,4,4,8)
    v = torch.from_numpy(g["step_v"])
    Bx, Tx, _, hx, wx = x.shape
    vtok = v.permute(0, 1, 3, 4, 2).reshape(-1, 4)
    V = dev_f(torch.cat([vtok, vtok]))
    xl = dev_f(x)
    for i in range(3):
        hip.cfg_euler_step(V, 4, xl, Bx, Tx, hx * wx, float(sig[i]), float(sig[i + 1]), 1.0, 3.0)
        check(xl, torch.from_numpy(g["step_x"][i + 1]), max_tol=2e-6, rms_tol=1e-6, what=f"euler step {i}")
    # guidance: v = vu + g_t (vc - vu)
    vu, vc = rnd(B * T * h * w, 4, seed=5), rnd(B * T * h * w, 4, seed=6)
    xl = dev_f(lat)
    hip.cfg_euler_step(dev_f(torch.cat([vu, vc])), 4, xl, B, T, h * w, 2.5, 1.25, 1.0, 3.0)
    gs = torch.linspace(1.0, 3.0, T).reshape(1, T, 1, 1, 1)
    to5 = lambda t: t.reshape(B, T, h, w, 4).permute(0, 1, 4, 2, 3)  # noqa: E731
    vv = to5(vu) + gs * (to5(vc) - to5(vu))
    check(xl, O.euler_step(vv, lat, 2.5, 1.25), max_tol=2e-6, rms_tol=1e-6, what="cfg + euler")


def test_emb_combine(hip):
    Bc, B, T, E = 4, 2, 3, 256
    t, a, n = rnd(Bc, E, seed=1), rnd(B * T, E, seed=2), rnd(Bc, E, seed=3)
    out = torch.empty(Bc * T, E, dtype=torch.bfloat16, device=DEV)
    hip.emb_combine(dev_f(t), dev_f(a), dev_f(n), Bc, B, T, E, out)
    ref = t.repeat_interleave(T, 0) + a.reshape(B, T, E).repeat(Bc // B, 1, 1).reshape(Bc * T, E) + n.repeat_interleave(T, 0)
    check(out, F.silu(ref), what="emb_combine")


# ----------------------------------------------------------------------------------------------
# the BIG tile configuration (256 x 160, 8 waves, 3-stage LDS ring) is selected for large M only
# ----------------------------------------------------------------------------------------------
def test_gemm_big_tile_dense_and_geglu(hip):
    from wiw_amd import hip as H
    from wiw_amd.unet import pack_geglu

    M, N, K = 49152 + 37, 320, 192      # ragged last M tile; 3 K tiles (prologue + steady state + drain)
    a, w = bf(rnd(M, K, seed=1)), bf(rnd(N, K, seed=2) / math.sqrt(K))
    bias, r1 = rnd(N, seed=3), bf(rnd(M, N, seed=5))
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    hip.gemm(dev_bf(a), dev_bf(w), out, M=M, N=N, K=K, C1=K, bias=dev_f(bias), res1=dev_bf(r1), ldr1=N, beta1=1.0)
    check(out, a @ w.t() + bias + r1, what="big-tile gemm + bias + residual")
    for K1 in (64, 128):  # 1 and 2 K tiles: shorter than the DMA pipeline depth
        hip.gemm(dev_bf(a[:, :K1].contiguous()), dev_bf(w[:, :K1].contiguous()), out, M=M, N=N, K=K1, C1=K1)
        check(out, a[:, :K1] @ w[:, :K1].t(), what=f"big-tile gemm K={K1}")
    Cn, Mg = 64, 24576 + 130
    x = bf(rnd(Mg, Cn, seed=6))
    wg, bg = rnd(8 * Cn, Cn, seed=7) / math.sqrt(Cn), rnd(8 * Cn, seed=8)
    wp, bp, _ = pack_geglu(wg, bg)
    og = torch.empty(Mg, 4 * Cn, dtype=torch.bfloat16, device=DEV)
    hip.gemm(dev_bf(x), dev_bf(wp), og, M=Mg, N=wp.shape[0], K=Cn, C1=Cn, bias=dev_f(bp), epilogue=H.EPI_GEGLU, n_out=4 * Cn)
    hh = x @ bf(wg).t() + bg
    val, gate = hh.chunk(2, dim=-1)
    check(og, val * F.gelu(gate), what="big-tile geglu")


def test_gemm_big_tile_conv_modes(hip):
    from wiw_amd import hip as H

    n, c, cout, h, w = 24, 64, 320, 32, 64          # M = 49152
    x = bf(rnd(n, c, h, w, seed=1))
    wt = bf(rnd(cout, c, 3, 3, seed=2) / math.sqrt(9 * c))
    b = rnd(cout, seed=3)
    wk = dev_bf(wt.permute(0, 2, 3, 1).reshape(cout, -1))
    xt = dev_bf(nhwc(x))
    M = n * h * w
    out = torch.empty(M, cout, dtype=torch.bfloat16, device=DEV)
    hip.gemm(xt, wk, out, M=M, N=cout, K=9 * c, C1=c, mode=H.A_CONV3X3, H=h, Wd=w, bias=dev_f(b))
    check(from_nhwc(out, n, h, w), F.conv2d(x, wt, b, padding=1), what="big-tile conv3x3")
    # temporal taps: B=4, T=6
    wtt = bf(rnd(cout, c, 3, 1, 1, seed=4) / math.sqrt(3 * c))
    wkt = dev_bf(wtt[:, :, :, 0, 0].permute(0, 2, 1).reshape(cout, -1))
    hip.gemm(xt, wkt, out, M=M, N=cout, K=3 * c, C1=c, mode=H.A_CONV_T3, H=h, Wd=w, T=6, bias=dev_f(b))
    x5 = x.reshape(4, 6, c, h, w).permute(0, 2, 1, 3, 4)
    ref = F.conv3d(x5, wtt, b, padding=(1, 0, 0)).permute(0, 2, 1, 3, 4).reshape(n, cout, h, w)
    check(from_nhwc(out, n, h, w), ref, what="big-tile temporal conv")
    # nearest-up + conv from a quarter-size input, and stride 2 from a 4x input
    xs = bf(rnd(n, c, h // 2, w // 2, seed=5))
    hip.gemm(dev_bf(nhwc(xs)), wk, out, M=M, N=cout, K=9 * c, C1=c, mode=H.A_CONV3X3_UP, H=h, Wd=w, bias=dev_f(b))
    ref = F.conv2d(F.interpolate(xs, scale_factor=2.0, mode="nearest"), wt, b, padding=1)
    check(from_nhwc(out, n, h, w), ref, what="big-tile up-conv")
    xl = bf(rnd(6, c, 2 * h, 2 * w * 2, seed=6))     # 6 frames of 64 x 256 -> output 32 x 128, M = 24576 ... x 2 N tiles
    Mo = 6 * h * (2 * w)
    o2 = torch.empty(Mo, 640, dtype=torch.bfloat16, device=DEV)
    wt2 = bf(rnd(640, c, 3, 3, seed=7) / math.sqrt(9 * c))
    hip.gemm(dev_bf(nhwc(xl)), dev_bf(wt2.permute(0, 2, 3, 1).reshape(640, -1)), o2, M=Mo, N=640, K=9 * c, C1=c,
             mode=H.A_CONV3X3_S2, H=h, Wd=2 * w)
    check(from_nhwc(o2, 6, h, 2 * w), F.conv2d(xl, wt2, None, stride=2, padding=1), what="big-tile stride-2 conv")


# ----------------------------------------------------------------------------------------------
# the HUGE tile configuration (256 x 320, 4 x 2 waves of 64 x 160, gemm_huge.hip) takes bf16-output GEMMs with
# N % 320 == 0 and >= 200 output tiles
# ----------------------------------------------------------------------------------------------
def test_gemm_huge_tile_dense(hip):
    M, N, K = 256 * 210 + 37, 320, 704      # ragged last M tile; 11 K tiles
    a, w = bf(rnd(M, K, seed=1)), bf(rnd(N, K, seed=2) / math.sqrt(K))
    bias, r1, r2 = rnd(N, seed=3), bf(rnd(M, N, seed=5)), bf(rnd(M, N, seed=6))
    rpv = 1024
    rowvec = rnd((M + rpv - 1) // rpv, N + 8, seed=4)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    hip.gemm(dev_bf(a), dev_bf(w), out, M=M, N=N, K=K, C1=K, bias=dev_f(bias), res1=dev_bf(r1), ldr1=N, beta1=1.0)
    check(out, a @ w.t() + bias + r1, what="huge-tile gemm + bias + residual")
    # the AlphaBlender form: alpha * (acc + bias + rowvec[frame]) + beta1 * res1 + beta2 * res2
    hip.gemm(dev_bf(a), dev_bf(w), out, M=M, N=N, K=K, C1=K, bias=dev_f(bias), rowvec=dev_f(rowvec), rowvec_ld=N + 8,
             rows_per_vec=rpv, alpha=0.3, res1=dev_bf(r1), ldr1=N, beta1=0.3, res2=dev_bf(r2), ldr2=N, beta2=0.7)
    rv = rowvec[:, :N].repeat_interleave(rpv, dim=0)[:M]
    check(out, 0.3 * (a @ w.t() + bias + rv) + 0.3 * r1 + 0.7 * r2, what="huge-tile gemm blend epilogue")
    for K1 in (640,):  # the shortest K the HUGE tile takes (10 K tiles)
        hip.gemm(dev_bf(a[:, :K1].contiguous()), dev_bf(w[:, :K1].contiguous()), out, M=M, N=N, K=K1, C1=K1)
        check(out, a[:, :K1] @ w[:, :K1].t(), what=f"huge-tile gemm K={K1}")
    # two N tiles (both wave columns of a second tile), concat input, back-to-back tiles per block
    M2, N2 = 256 * 110 + 8, 640
    a1, a2 = bf(rnd(M2, 320, seed=7)), bf(rnd(M2, 384, seed=8))
    w2 = bf(rnd(N2, 704, seed=9) / math.sqrt(704))
    o2 = torch.empty(M2, N2, dtype=torch.bfloat16, device=DEV)
    hip.gemm(dev_bf(a1), dev_bf(w2), o2, M=M2, N=N2, K=704, C1=320, A2=dev_bf(a2), C2=384)
    check(o2, torch.cat([a1, a2], dim=1) @ w2.t(), what="huge-tile gemm, 2 N tiles, concat A")


def test_gemm_huge_tile_geglu(hip):
    from wiw_amd import hip as H
    from wiw_amd.unet import pack_geglu

    Cn, Mg = 640, 256 * 25 + 130           # N packed = 5120 -> 16 N tiles x 26 M tiles
    x = bf(rnd(Mg, Cn, seed=6))
    wg, bg = rnd(8 * Cn, Cn, seed=7) / math.sqrt(Cn), rnd(8 * Cn, seed=8)
    wp, bp, _ = pack_geglu(wg, bg)
    assert wp.shape[0] % 320 == 0
    og = torch.empty(Mg, 4 * Cn, dtype=torch.bfloat16, device=DEV)
    hip.gemm(dev_bf(x), dev_bf(wp), og, M=Mg, N=wp.shape[0], K=Cn, C1=Cn, bias=dev_f(bp), epilogue=H.EPI_GEGLU, n_out=4 * Cn)
    hh = x @ bf(wg).t() + bg
    val, gate = hh.chunk(2, dim=-1)
    check(og, val * F.gelu(gate), what="huge-tile geglu")


def test_gemm_huge_tile_conv_modes(hip):
    from wiw_amd import hip as H

    n, c, cout, h, w = 28, 256, 320, 32, 64         # M = 57344 -> 224 tiles; K = 2304 (3x3), 768 (temporal)
    x = bf(rnd(n, c, h, w, seed=1))
    wt = bf(rnd(cout, c, 3, 3, seed=2) / math.sqrt(9 * c))
    b = rnd(cout, seed=3)
    wk = dev_bf(wt.permute(0, 2, 3, 1).reshape(cout, -1))
    xt = dev_bf(nhwc(x))
    M = n * h * w
    out = torch.empty(M, cout, dtype=torch.bfloat16, device=DEV)
    hip.gemm(xt, wk, out, M=M, N=cout, K=9 * c, C1=c, mode=H.A_CONV3X3, H=h, Wd=w, bias=dev_f(b))
    check(from_nhwc(out, n, h, w), F.conv2d(x, wt, b, padding=1), what="huge-tile conv3x3")
    wtt = bf(rnd(cout, c, 3, 1, 1, seed=4) / math.sqrt(3 * c))
    wkt = dev_bf(wtt[:, :, :, 0, 0].permute(0, 2, 1).reshape(cout, -1))
    hip.gemm(xt, wkt, out, M=M, N=cout, K=3 * c, C1=c, mode=H.A_CONV_T3, H=h, Wd=w, T=7, bias=dev_f(b))
    x5 = x.reshape(4, 7, c, h, w).permute(0, 2, 1, 3, 4)
    ref = F.conv3d(x5, wtt, b, padding=(1, 0, 0)).permute(0, 2, 1, 3, 4).reshape(n, cout, h, w)
    check(from_nhwc(out, n, h, w), ref, what="huge-tile temporal conv")
    xs = bf(rnd(n, c, h // 2, w // 2, seed=5))
    hip.gemm(dev_bf(nhwc(xs)), wk, out, M=M, N=cout, K=9 * c, C1=c, mode=H.A_CONV3X3_UP, H=h, Wd=w, bias=dev_f(b))
    ref = F.conv2d(F.interpolate(xs, scale_factor=2.0, mode="nearest"), wt, b, padding=1)
    check(from_nhwc(out, n, h, w), ref, what="huge-tile up-conv")
    xl = bf(rnd(7, c, 2 * h, 4 * w, seed=6))         # 7 frames of 64 x 256 -> 32 x 128 outputs, M = 28672, 2 N tiles
    Mo = 7 * h * (2 * w)
    o2 = torch.empty(Mo, 640, dtype=torch.bfloat16, device=DEV)
    wt2 = bf(rnd(640, c, 3, 3, seed=7) / math.sqrt(9 * c))
    wk2 = dev_bf(wt2.permute(0, 2, 3, 1).reshape(640, -1))
    hip.gemm(dev_bf(nhwc(xl)), wk2, o2, M=Mo, N=640, K=9 * c, C1=c, mode=H.A_CONV3X3_S2, H=h, Wd=2 * w)
    check(from_nhwc(o2, 7, h, 2 * w), F.conv2d(xl, wt2, None, stride=2, padding=1), what="huge-tile stride-2 conv")
    hip.gemm(dev_bf(nhwc(xl)), wk2, o2, M=Mo, N=640, K=9 * c, C1=c, mode=H.A_CONV3X3_S2P, H=h, Wd=2 * w)
    check(from_nhwc(o2, 7, h, 2 * w), F.conv2d(F.pad(xl, (0, 1, 0, 1)), wt2, None, stride=2), what="huge-tile stride-2 (0,1,0,1) conv")


def test_gemm_huge_tile_partial_n(hip):
    """N = 512 (one full + one 192-wide 320-column tile): the VAE decoder's channel counts on the 256 x 320 tile."""
    from wiw_amd import hip as H

    n, c, cout, h, w = 28, 128, 512, 32, 64         # M = 57344 -> 224 x 2 tiles, K = 1152
    x = bf(rnd(n, c, h, w, seed=1))
    wt = bf(rnd(cout, c, 3, 3, seed=2) / math.sqrt(9 * c))
    b, r1 = rnd(cout, seed=3), bf(rnd(n * h * w, cout, seed=4))
    M = n * h * w
    out = torch.empty(M, cout, dtype=torch.bfloat16, device=DEV)
    hip.gemm(dev_bf(nhwc(x)), dev_bf(wt.permute(0, 2, 3, 1).reshape(cout, -1)), out, M=M, N=cout, K=9 * c, C1=c,
             mode=H.A_CONV3X3, H=h, Wd=w, bias=dev_f(b), res1=dev_bf(r1), ldr1=cout, beta1=1.0)
    ref = F.conv2d(x, wt, b, padding=1) + from_nhwc(r1, n, h, w)
    check(from_nhwc(out, n, h, w), ref, what="huge-tile conv3x3, N = 512 (partial N tile)")


@pytest.mark.parametrize("n,c,c2,c3,cout,h,w", [(3, 64, 64, 0, 96, 8, 16),          # small tile, single-source shortcut
                                                 (24, 64, 128, 64, 320, 32, 64),    # 256 x 160 tile, concat shortcut
                                                 (28, 128, 64, 192, 320, 32, 64)])  # 256 x 320 tile (K = 1408)
def test_conv3x3_with_fused_shortcut(hip, n, c, c2, c3, cout, h, w):
    """conv2 + the 1x1 conv_shortcut over the skip concat (resnet.py:311-318) as ONE implicit GEMM: K = 9*C1 + C2 + C3."""
    from wiw_amd import hip as H

    x = bf(rnd(n, c, h, w, seed=1))
    s1 = bf(rnd(n, c2, h, w, seed=2))
    s2 = bf(rnd(n, c3, h, w, seed=3)) if c3 else None
    wt = bf(rnd(cout, c, 3, 3, seed=4) / math.sqrt(9 * c))
    wsc = bf(rnd(cout, c2 + c3, 1, 1, seed=5) / math.sqrt(c2 + c3))
    b = rnd(cout, seed=6)
    wk = dev_bf(torch.cat([wt.permute(0, 2, 3, 1).reshape(cout, -1), wsc[:, :, 0, 0]], dim=1))
    M = n * h * w
    out = torch.empty(M, cout, dtype=torch.bfloat16, device=DEV)
    hip.gemm(dev_bf(nhwc(x)), wk, out, M=M, N=cout, K=9 * c + c2 + c3, C1=c, mode=H.A_CONV3X3, H=h, Wd=w,
             A2=dev_bf(nhwc(s1)), C2=c2, A3=dev_bf(nhwc(s2)) if c3 else None, C3=c3, bias=dev_f(b))
    skip = torch.cat([s1, s2], dim=1) if c3 else s1
    ref = F.conv2d(x, wt, b, padding=1) + F.conv2d(skip, wsc)
    check(from_nhwc(out, n, h, w), ref, what=f"conv3x3 + fused shortcut {c}|{c2}+{c3}->{cout}")


def test_attn_spatial_long_sequence_many_blocks(hip):
    """A level-1-sized problem: 36 KV tiles per block (lazy rescale in steady state), 1152 blocks over the XCD remap."""
    frames, S, heads = 8, 2304, 8
    C = heads * 64
    qkv = bf(rnd(frames * S, 3 * C, seed=21))
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    qk = dev_bf(torch.cat([q, k], dim=1))
    vt = dev_bf(v.t().contiguous())
    o = torch.empty(frames * S, C, dtype=torch.bfloat16, device=DEV)
    hip.attn_spatial(qk, 2 * C, C, vt, frames * S, o, C, frames, S, heads, 0.125)
    ref = sdpa(q.reshape(frames, S, C), k.reshape(frames, S, C), v.reshape(frames, S, C), heads).reshape(frames * S, C)
    check(o, ref, what="attn_spatial 8 x 2304 x 8 heads")


# ----------------------------------------------------------------------------------------------
# tiled weights (hip.TiledW): same LDS image, same bits
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K,mode,epi", [(300, 320, 320, 0, 0), (4032, 1280, 1280, 0, 0), (130, 4, 64, 0, 0), (1000, 200, 128, 0, 0),
                                            (128, 13, 192, 0, 0), (520, 2560, 320, 0, 1), (16384, 640, 2560, 0, 0),
                                            (2 * 18 * 32, 320, 9 * 128, 1, 0), (70000, 320, 9 * 64, 1, 0), (2 * 4 * 64, 128, 3 * 128, 4, 0)])
def test_gemm_tiled_weight_is_bit_identical(hip, M, N, K, mode, epi):
    """The tiled layout only changes WHERE the LDS-DMA reads W from; every output bit must equal the row-major run."""
    from wiw_amd import hip as H

    taps = {0: 1, 1: 9, 4: 3}[mode]
    C1 = K // taps
    a = dev_bf(rnd(M, C1, seed=1))
    w = dev_bf(rnd(N, K, seed=2) / math.sqrt(K))
    bias = dev_f(rnd(N, seed=3))
    kw = dict(M=M, N=N, K=K, C1=C1, mode=mode, bias=bias, epilogue=epi)
    if mode == 1:
        kw.update(H=18 if M == 2 * 18 * 32 else 250, Wd=32 if M == 2 * 18 * 32 else 280)
    if mode == 4:
        kw.update(H=8, Wd=8, T=4)
    if epi & H.EPI_GEGLU:
        kw["n_out"] = N // 2
    outs = []
    for W in (w, H.TiledW(w)):
        out = torch.full((M, N // 2 if epi & H.EPI_GEGLU else N), float("nan"), dtype=torch.bfloat16, device=DEV)
        hip.gemm(a, W, out, **kw)
        outs.append(out)
    torch.cuda.synchronize()
    assert torch.isfinite(outs[0].float()).all()
    assert torch.equal(outs[0], outs[1]), f"tiled W changed the result (M={M} N={N} K={K} mode={mode} epi={epi})"


# ----------------------------------------------------------------------------------------------
# split-K (WiwGemmArgs.splitk): K ranges as extra schedule rows, fp32 partial slabs, reduce + epilogue kernel
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("sk", [2, 4])
@pytest.mark.parametrize("tiled", [False, True])
def test_gemm_splitk_dense_all_epilogue_terms(hip, sk, tiled):
    from wiw_amd import hip as H

    M, N, K = 1000, 320, 2048            # 4 x 2 tiles of 256 x 160, partial last M tile, 32 K tiles
    a, w = bf(rnd(M, K, seed=1)), bf(rnd(N, K, seed=2) / math.sqrt(K))
    b, rv = rnd(N, seed=3), rnd(M // 100, 2 * N, seed=4)
    r1, r2 = bf(rnd(M, N, seed=5)), bf(rnd(M, N, seed=6))
    W = H.TiledW(dev_bf(w)) if tiled else dev_bf(w)
    kw = dict(M=M, N=N, K=K, C1=K, bias=dev_f(b), rowvec=dev_f(rv)[:, N // 2:], rowvec_ld=2 * N, rows_per_vec=100,
              alpha=0.75, res1=dev_bf(r1), ldr1=N, beta1=0.5, res2=dev_bf(r2), ldr2=N, beta2=-0.25)
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    hip.gemm(dev_bf(a), W, out, splitk=sk, **kw)
    rows = torch.arange(M) // 100
    ref = 0.75 * (a @ w.t() + b + rv[rows][:, N // 2:N // 2 + N]) + 0.5 * r1 - 0.25 * r2
    check(out, ref, what=f"split-K {sk} dense, all epilogue terms, tiled={tiled}")
    one = torch.empty_like(out)
    hip.gemm(dev_bf(a), W, one, **kw)
    check(out, one.float(), max_tol=1.2e-2, rms_tol=3e-3, what="split-K vs single pass")   # two independent bf16 roundings
    again = torch.empty_like(out)
    hip.gemm(dev_bf(a), W, again, splitk=sk, **kw)
    assert torch.equal(out, again), "split-K must be deterministic (slabs are added in range order)"
    f32 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    hip.gemm(dev_bf(a), W, f32, M=M, N=N, K=K, C1=K, bias=dev_f(b), epilogue=H.EPI_OUT_F32 | H.EPI_SILU, splitk=sk)
    check(f32, F.silu(a @ w.t() + b), max_tol=1e-4, rms_tol=1e-5, what="split-K fp32 + SiLU")


@pytest.mark.parametrize("sk,c,c2,c3", [(2, 128, 64, 64), (3, 64, 64, 128), (2, 64, 384, 320), (4, 256, 0, 0)])
def test_gemm_splitk_conv3x3_with_shortcut_segment(hip, sk, c, c2, c3):
    """K ranges that start at a tap, in the middle of a tap (c = 256: 4.5 taps per range) and INSIDE the fused-shortcut
    segment (c = 64 with a 704-channel concat: the second range starts one K tile into the segment)."""
    from wiw_amd import hip as H

    n, h, w, cout = 3, 12, 16, 320
    x = bf(rnd(n, c, h, w, seed=1))
    wt = bf(rnd(cout, c, 3, 3, seed=4) / math.sqrt(9 * c))
    b = rnd(cout, seed=6)
    M = n * h * w
    out = torch.empty(M, cout, dtype=torch.bfloat16, device=DEV)
    if c2:
        s1, s2 = bf(rnd(n, c2, h, w, seed=2)), bf(rnd(n, c3, h, w, seed=3))
        wsc = bf(rnd(cout, c2 + c3, 1, 1, seed=5) / math.sqrt(c2 + c3))
        wk = dev_bf(torch.cat([wt.permute(0, 2, 3, 1).reshape(cout, -1), wsc[:, :, 0, 0]], dim=1))
        hip.gemm(dev_bf(nhwc(x)), wk, out, M=M, N=cout, K=9 * c + c2 + c3, C1=c, mode=H.A_CONV3X3, H=h, Wd=w,
                 A2=dev_bf(nhwc(s1)), C2=c2, A3=dev_bf(nhwc(s2)), C3=c3, bias=dev_f(b), splitk=sk)
        ref = F.conv2d(x, wt, b, padding=1) + F.conv2d(torch.cat([s1, s2], dim=1), wsc)
    else:
        wk = dev_bf(wt.permute(0, 2, 3, 1).reshape(cout, -1))
        K = 9 * c
        hip.gemm(dev_bf(nhwc(x)), wk, out, M=M, N=cout, K=K, C1=c, mode=H.A_CONV3X3, H=h, Wd=w, bias=dev_f(b), splitk=sk)
        ref = F.conv2d(x, wt, b, padding=1)
    check(from_nhwc(out, n, h, w), ref, what=f"split-K {sk} conv3x3 {c}|{c2}+{c3}")


def test_gemm_splitk_conv_temporal_and_validation(hip):
    from wiw_amd import hip as H

    B, T, c, h, w = 2, 6, 128, 4, 8
    x = bf(rnd(B, c, T, h, w, seed=1))
    wt = bf(rnd(c, c, 3, 1, 1, seed=2) / math.sqrt(3 * c))
    b = rnd(c, seed=3)
    tok = x.permute(0, 2, 3, 4, 1).reshape(-1, c)
    M = tok.shape[0]
    out = torch.empty(M, c, dtype=torch.bfloat16, device=DEV)
    hip.gemm(dev_bf(tok), dev_bf(wt[:, :, :, 0, 0].permute(0, 2, 1).reshape(c, -1)), out, M=M, N=c, K=3 * c, C1=c,
             mode=H.A_CONV_T3, H=h, Wd=w, T=T, bias=dev_f(b), splitk=2)
    ref = F.conv3d(x, wt, b, padding=(1, 0, 0)).permute(0, 2, 3, 4, 1).reshape(-1, c)
    check(out, ref, what="split-K 2 temporal conv")
    with pytest.raises(RuntimeError, match="divisible"):
        hip.gemm(dev_bf(tok), dev_bf(wt[:, :, :, 0, 0].permute(0, 2, 1).reshape(c, -1)), out, M=M, N=c, K=3 * c, C1=c,
                 mode=H.A_CONV_T3, H=h, Wd=w, T=T, splitk=4)


def test_gemm_splitk_on_the_256x320_tile(hip):
    """Shapes whose K ranges the launcher puts on gemm_huge.hip (N % 320 == 0, >= 10 K tiles per range, >= 200 items):
    dense with every epilogue term, and a conv3x3 + fused shortcut whose ranges start in the middle of a tap."""
    from wiw_amd import hip as H

    M, N, K, sk = 6400, 640, 2560, 4
    a, w = bf(rnd(M, K, seed=1)), bf(rnd(N, K, seed=2) / math.sqrt(K))
    b, r1 = rnd(N, seed=3), bf(rnd(M, N, seed=5))
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    hip.gemm(dev_bf(a), H.TiledW(dev_bf(w)), out, M=M, N=N, K=K, C1=K, bias=dev_f(b), alpha=0.5, res1=dev_bf(r1), ldr1=N,
             beta1=1.0, splitk=sk)
    check(out, 0.5 * (a @ w.t() + b) + r1, what="split-K 4 dense on the 256x320 tile")

    n, h, w_, c, c2, c3, cout, sk = 4, 40, 56, 256, 128, 64, 640, 3
    x = bf(rnd(n, c, h, w_, seed=1))
    s1, s2 = bf(rnd(n, c2, h, w_, seed=2)), bf(rnd(n, c3, h, w_, seed=3))
    wt = bf(rnd(cout, c, 3, 3, seed=4) / math.sqrt(9 * c))
    wsc = bf(rnd(cout, c2 + c3, 1, 1, seed=5) / math.sqrt(c2 + c3))
    bb = rnd(cout, seed=6)
    wk = H.TiledW(dev_bf(torch.cat([wt.permute(0, 2, 3, 1).reshape(cout, -1), wsc[:, :, 0, 0]], dim=1)))
    M = n * h * w_
    out = torch.full((M, cout), float("nan"), dtype=torch.bfloat16, device=DEV)
    hip.gemm(dev_bf(nhwc(x)), wk, out, M=M, N=cout, K=9 * c + c2 + c3, C1=c, mode=H.A_CONV3X3, H=h, Wd=w_,
             A2=dev_bf(nhwc(s1)), C2=c2, A3=dev_bf(nhwc(s2)), C3=c3, bias=dev_f(bb), splitk=sk)
    ref = F.conv2d(x, wt, bb, padding=1) + F.conv2d(torch.cat([s1, s2], dim=1), wsc)
    check(from_nhwc(out, n, h, w_), ref, what="split-K 3 conv3x3 + shortcut on the 256x320 tile")


# ----------------------------------------------------------------------------------------------
# LayerNorm folded into the consumer GEMM (WIW_EPI_LNFOLD): x raw, W = W * gamma, [s | t], statistics in the kernel
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,C,N", [(520, 320, 960), (70000, 320, 960), (1000, 64, 320), (300, 128, 160)])
def test_gemm_layernorm_fold_plain(hip, M, C, N):
    """Plain staged epilogue (the q|k|v projection after norm1), with a residual and a per-row-group vector on top;
    small M takes the 128x160 tile, M = 70000 the 256x160 tile.  The only extra error over the two-kernel form is the
    16-bit rounding of W * gamma (vs rounding LN(x)): gate 1.5e-2 / 5e-3."""
    from wiw_amd import hip as H
    from wiw_amd.unet import fold_layernorm

    x = bf(rnd(M, C, seed=1) * 1.5 + 0.4)
    x[:, 5] += 6.0                                              # a channel with a large offset: the fold must remove the mean
    w, b = rnd(N, C, seed=2) / math.sqrt(C), rnd(N, seed=3) * 0.2
    gamma, beta = 1 + 0.3 * rnd(C, seed=4), 0.2 * rnd(C, seed=5)
    r1 = bf(rnd(M, N, seed=6))
    rv = rnd(-(-M // 100), N, seed=7)
    wg, st = fold_layernorm(w, b, gamma, beta, torch.bfloat16)
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
    hip.gemm(dev_bf(x), H.TiledW(wg.to(DEV)), out, M=M, N=N, K=C, C1=C, lnfold=st.to(DEV), ln_eps=1e-5,
             res1=dev_bf(r1), ldr1=N, beta1=1.0, rowvec=dev_f(rv), rowvec_ld=N, rows_per_vec=100)
    ref = F.layer_norm(x, (C,), gamma, beta, 1e-5) @ w.t() + b + r1 + rv[torch.arange(M) // 100]
    check(out, ref, max_tol=1.5e-2, rms_tol=5e-3, what=f"LN-fold GEMM {M}x{N}x{C}")
    again = torch.empty_like(out)
    hip.gemm(dev_bf(x), H.TiledW(wg.to(DEV)), again, M=M, N=N, K=C, C1=C, lnfold=st.to(DEV), ln_eps=1e-5,
             res1=dev_bf(r1), ldr1=N, beta1=1.0, rowvec=dev_f(rv), rowvec_ld=N, rows_per_vec=100)
    assert torch.equal(out, again)
    # a row's result does not depend on which tile / block computed it: the first 300 rows alone give the same bits
    part = torch.empty(300, N, dtype=torch.bfloat16, device=DEV)
    hip.gemm(dev_bf(x[:300]), H.TiledW(wg.to(DEV)), part, M=300, N=N, K=C, C1=C, lnfold=st.to(DEV), ln_eps=1e-5,
             res1=dev_bf(r1[:300]), ldr1=N, beta1=1.0, rowvec=dev_f(rv), rowvec_ld=N, rows_per_vec=100)
    assert torch.equal(part, out[:300])


@pytest.mark.parametrize("M,C", [(520, 320), (66000, 320), (777, 64)])
def test_gemm_layernorm_fold_geglu(hip, M, C):
    """GEGLU projection after norm3 / norm_in: gamma-folded packed weights, s and t in the packed row order."""
    from wiw_amd import hip as H
    from wiw_amd.unet import pack_geglu

    x = bf(rnd(M, C, seed=1) * 1.2 - 0.3)
    x[:, 9] -= 5.0
    w0, b0 = rnd(8 * C, C, seed=2) / math.sqrt(C), rnd(8 * C, seed=3) * 0.2
    gamma, beta = 1 + 0.3 * rnd(C, seed=4), 0.2 * rnd(C, seed=5)
    wgp, tp, n_half = pack_geglu(w0 * gamma[None, :], w0 @ beta + b0)
    wgp = wgp.to(torch.bfloat16)
    st = torch.stack([wgp.float().sum(dim=1), tp.float()]).contiguous()
    out = torch.full((M, n_half), float("nan"), dtype=torch.bfloat16, device=DEV)
    hip.gemm(dev_bf(x), H.TiledW(wgp.to(DEV)), out, M=M, N=wgp.shape[0], K=C, C1=C, epilogue=H.EPI_GEGLU, n_out=n_half,
             lnfold=st.to(DEV), ln_eps=1e-5)
    y = F.layer_norm(x, (C,), gamma, beta, 1e-5) @ w0.t() + b0
    check(out, y[:, :n_half] * F.gelu(y[:, n_half:]), max_tol=1.5e-2, rms_tol=5e-3, what=f"LN-fold GEGLU {M}x{C}")


def test_gemm_layernorm_fold_validation(hip):
    from wiw_amd import hip as H

    a = dev_bf(rnd(128, 64, seed=1)); w = dev_bf(rnd(160, 64, seed=2)); st = dev_f(rnd(2, 160, seed=3))
    out = torch.empty(128, 160, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match="bias in t"):
        hip.gemm(a, w, out, M=128, N=160, K=64, C1=64, lnfold=st, bias=dev_f(rnd(160)))
    with pytest.raises(RuntimeError, match="N % 160"):
        hip.gemm(a, dev_bf(rnd(128, 64, seed=2)), torch.empty(128, 128, dtype=torch.bfloat16, device=DEV), M=128, N=128, K=64,
                 C1=64, lnfold=dev_f(rnd(2, 128)))


@pytest.mark.gpu
@pytest.mark.parametrize("rows,C,c0,ldx,pad", [(64, 64, 0, 64, 0), (200, 72, 8, 96, 8), (4032, 320, 640, 960, 64), (8, 8, 0, 8, 0),
                                               (1000, 136, 0, 136, 24)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_transpose_is_exact(rows, C, c0, ldx, pad, dtype):
    """wiw_transpose_bf16: Y[c][r] = X[r][c0 + c], bit for bit — whole and ragged 64 x 64 tiles, a column window of a wider
    matrix (the V block of a fused q|k|v projection), a padded output row stride (the zero-padded K of the weight-gradient
    GEMMs keeps its zeros), both 16-bit types (a 16-bit move: the same kernel serves both)."""
    from wiw_amd.hip import Hip

    hip = Hip(torch.device(DEV), dtype)
    g = torch.Generator().manual_seed(rows + C)
    X = torch.randn(rows, ldx, generator=g).to(dtype).to(DEV)
    ldy = rows + pad
    Y = torch.full((C, ldy), 7.0, dtype=dtype, device=DEV)
    hip.transpose(X, ldx, c0, rows, C, Y, ldy)
    assert torch.equal(Y[:, :rows], X[:, c0:c0 + C].t())
    assert bool((Y[:, rows:] == 7.0).all())                                    # nothing written past the rows


# ----------------------------------------------------------------------------------------------
# WIW_K_CMAJOR (round 4): the K index of the convolution modes in channel-block-major order
# ----------------------------------------------------------------------------------------------
def test_conv_modes_with_channel_block_major_k(hip):
    """k = ((c / 64) * taps + tap) * 64 + c % 64 (include/wiw_svd.h): a permutation of K on both operands — every conv mode,
    the fused shortcut segment behind the taps, split-K ranges that start inside a channel block / inside the segment, on
    the 256x320 tile (the large shapes) and the 128x160 tile (the small ones).  Against fp32 torch, and against the tap-major launch
    of the same problem (the two differ by the fp32 summation order only)."""
    from wiw_amd import hip as H
    from wiw_amd.unet import conv_k_cmajor

    # 3x3, 3x3 + shortcut, stride 2, upsample — small and 256x320-sized
    for (n, c, c2, c3, cout, h, w, sk) in [(3, 128, 0, 0, 64, 8, 16, 1), (3, 128, 64, 128, 320, 12, 16, 1),
                                           (4, 256, 128, 64, 640, 40, 56, 1), (3, 64, 384, 320, 320, 12, 16, 2),
                                           (3, 256, 0, 0, 320, 12, 16, 4), (4, 256, 128, 64, 640, 40, 56, 3)]:
        x = bf(rnd(n, c, h, w, seed=1))
        wt = bf(rnd(cout, c, 3, 3, seed=4) / math.sqrt(9 * c))
        b = rnd(cout, seed=6)
        M = n * h * w
        w_tap = wt.permute(0, 2, 3, 1).reshape(cout, -1)
        kw = dict(M=M, N=cout, C1=c, mode=H.A_CONV3X3, H=h, Wd=w, bias=dev_f(b), splitk=sk)
        ref = F.conv2d(x, wt, b, padding=1)
        w_cm, w_tm = conv_k_cmajor(w_tap, 9), w_tap
        if c2:
            s1, s2 = bf(rnd(n, c2, h, w, seed=2)), bf(rnd(n, c3, h, w, seed=3))
            wsc = bf(rnd(cout, c2 + c3, 1, 1, seed=5) / math.sqrt(c2 + c3))[:, :, 0, 0]
            w_cm, w_tm = torch.cat([w_cm, wsc], 1), torch.cat([w_tm, wsc], 1)
            kw.update(A2=dev_bf(nhwc(s1)), C2=c2, A3=dev_bf(nhwc(s2)), C3=c3)
            ref = ref + F.conv2d(torch.cat([s1, s2], dim=1), wsc[:, :, None, None])
        K = w_cm.shape[1]
        o_cm = torch.empty(M, cout, dtype=torch.bfloat16, device=DEV)
        o_tm = torch.empty_like(o_cm)
        hip.gemm(dev_bf(nhwc(x)), H.TiledW(dev_bf(w_cm)), o_cm, K=K, epilogue=H.K_CMAJOR, **kw)
        hip.gemm(dev_bf(nhwc(x)), H.TiledW(dev_bf(w_tm)), o_tm, K=K, **kw)
        check(from_nhwc(o_cm, n, h, w), ref, what=f"K-cmajor conv3x3 {c}|{c2}+{c3}->{cout} splitk={sk}")
        d = (o_cm.float() - o_tm.float()).cpu()
        rms = float(d.pow(2).mean().sqrt() / o_tm.float().pow(2).mean().sqrt().cpu())
        assert rms <= 2e-3, "channel-block-major and tap-major launches differ by more than the summation order"
    n, c, h, w = 3, 128, 8, 16
    x = bf(rnd(n, c, h, w, seed=1))
    wt = bf(rnd(c, c, 3, 3, seed=2) / math.sqrt(9 * c))
    b = rnd(c, seed=3)
    wk = H.TiledW(dev_bf(conv_k_cmajor(wt.permute(0, 2, 3, 1).reshape(c, -1), 9)))
    Mo = n * (h // 2) * (w // 2)
    out = torch.empty(Mo, c, dtype=torch.bfloat16, device=DEV)
    hip.gemm(dev_bf(nhwc(x)), wk, out, M=Mo, N=c, K=9 * c, C1=c, mode=H.A_CONV3X3_S2, H=h // 2, Wd=w // 2, bias=dev_f(b),
             epilogue=H.K_CMAJOR)
    check(from_nhwc(out, n, h // 2, w // 2), F.conv2d(x, wt, b, stride=2, padding=1), what="K-cmajor conv3x3 stride 2")
    Mo = n * 4 * h * w
    out = torch.empty(Mo, c, dtype=torch.bfloat16, device=DEV)
    hip.gemm(dev_bf(nhwc(x)), wk, out, M=Mo, N=c, K=9 * c, C1=c, mode=H.A_CONV3X3_UP, H=2 * h, Wd=2 * w, bias=dev_f(b),
             epilogue=H.K_CMAJOR)
    check(from_nhwc(out, n, 2 * h, 2 * w), F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), wt, b, padding=1),
          what="K-cmajor nearest-up + conv3x3")
    # temporal (3 taps), plain and split-K
    B, T, c, h, w = 2, 6, 128, 4, 8
    x5 = bf(rnd(B, c, T, h, w, seed=1))
    wt3 = bf(rnd(c, c, 3, 1, 1, seed=2) / math.sqrt(3 * c))
    tok = x5.permute(0, 2, 3, 4, 1).reshape(-1, c)
    M = tok.shape[0]
    ref = F.conv3d(x5, wt3, b, padding=(1, 0, 0)).permute(0, 2, 3, 4, 1).reshape(-1, c)
    wk3 = H.TiledW(dev_bf(conv_k_cmajor(wt3[:, :, :, 0, 0].permute(0, 2, 1).reshape(c, -1), 3)))
    for sk in (1, 2, 3):
        out = torch.empty(M, c, dtype=torch.bfloat16, device=DEV)
        hip.gemm(dev_bf(tok), wk3, out, M=M, N=c, K=3 * c, C1=c, mode=H.A_CONV_T3, H=h, Wd=w, T=T, bias=dev_f(b), splitk=sk,
                 epilogue=H.K_CMAJOR)
        check(out, ref, what=f"K-cmajor temporal conv splitk={sk}")
    with pytest.raises(RuntimeError, match="convolution-mode"):
        hip.gemm(dev_bf(tok), wk3, out, M=M, N=c, K=3 * c, C1=3 * c, epilogue=H.K_CMAJOR)


@pytest.mark.gpu
def test_conv3x3_halo_staged_kernel(hip):
    """WIW_K_HALO32 (include/wiw_svd.h): the 3x3 convolution whose A operand is a halo image per 32-channel block, read at nine
    pixel offsets.  Against fp32 torch and against the per-tap launch of the same problem (summation order only), at both
    widths the kernel takes, one and several tiles per frame, frames that start inside a grid round, one and two N tiles,
    1 / 3 / 10 channel-block pairs, with bias, the per-frame vector, a residual and alpha; the image border (zero padding)
    is every tile's first / last row and column.  Geometry outside the kernel's is refused, not re-routed."""
    from wiw_amd import hip as H
    from wiw_amd.unet import conv_k_cmajor, conv_k_halo32

    # (frames, C, C2, C3, Cout, H, W, vector + residual + alpha): 18 x 32 and 10 x 32 put frame edges INSIDE tiles (a zero row
    # between the frames); C2 / C3: the fused 1x1 shortcut segment behind the taps
    for (n, c, c2, c3, cout, h, w, extras) in [(2, 64, 0, 0, 320, 4, 64, False), (3, 192, 0, 0, 320, 2, 128, True),
                                               (2, 128, 0, 0, 640, 8, 64, True), (5, 64, 0, 0, 320, 12, 64, False),
                                               (28, 320, 0, 0, 320, 36, 64, True), (3, 640, 0, 0, 640, 4, 128, True),
                                               (4, 128, 0, 0, 320, 18, 32, True), (8, 64, 0, 0, 640, 10, 32, False),
                                               (28, 256, 0, 0, 1280, 18, 32, True), (4, 64, 128, 0, 320, 8, 64, True),
                                               (3, 128, 64, 192, 640, 4, 128, False), (4, 192, 128, 64, 320, 18, 32, True)]:
        x = bf(rnd(n, c, h, w, seed=1))
        wt = bf(rnd(cout, c, 3, 3, seed=4) / math.sqrt(9 * c))
        b = rnd(cout, seed=6)
        M = n * h * w
        assert H.Hip.conv_halo_ok(M, cout, c, h, w)
        w_tap = wt.permute(0, 2, 3, 1).reshape(cout, -1)
        kw = dict(M=M, N=cout, K=9 * c + c2 + c3, C1=c, mode=H.A_CONV3X3, H=h, Wd=w, bias=dev_f(b))
        ref = F.conv2d(x, wt, b, padding=1)
        w_h, w_c = conv_k_halo32(w_tap), conv_k_cmajor(w_tap, 9)
        if c2:
            s1 = bf(rnd(n, c2, h, w, seed=2))
            wsc = bf(rnd(cout, c2 + c3, 1, 1, seed=5) / math.sqrt(c2 + c3))[:, :, 0, 0]
            w_h, w_c = torch.cat([w_h, wsc], 1), torch.cat([w_c, wsc], 1)
            kw.update(A2=dev_bf(nhwc(s1)), C2=c2)
            src = s1
            if c3:
                s2 = bf(rnd(n, c3, h, w, seed=3))
                kw.update(A3=dev_bf(nhwc(s2)), C3=c3)
                src = torch.cat([s1, s2], dim=1)
            ref = ref + F.conv2d(src, wsc[:, :, None, None])
        if extras:
            vec = rnd(n, cout, seed=7)
            res = bf(rnd(n, cout, h, w, seed=8))
            kw.update(rowvec=dev_f(vec), rowvec_ld=cout, rows_per_vec=h * w, res1=dev_bf(nhwc(res)), ldr1=cout, beta1=1.0, alpha=0.75)
            ref = 0.75 * (ref + vec[:, :, None, None]) + res
        o_h = torch.empty(M, cout, dtype=torch.bfloat16, device=DEV)
        o_t = torch.empty_like(o_h)
        hip.gemm(dev_bf(nhwc(x)), H.TiledW(dev_bf(w_h)), o_h, epilogue=H.K_HALO32, **kw)
        hip.gemm(dev_bf(nhwc(x)), H.TiledW(dev_bf(w_c)), o_t, epilogue=H.K_CMAJOR, **kw)
        check(from_nhwc(o_h, n, h, w), ref, what=f"halo conv3x3 {c}|{c2}+{c3}->{cout} {n}x{h}x{w}")
        d = (o_h.float() - o_t.float()).cpu()
        rms = float(d.pow(2).mean().sqrt() / o_t.float().pow(2).mean().sqrt().cpu())
        print(f"[halo] conv3x3 {c}|{c2}+{c3}->{cout} {n}x{h}x{w}: rms vs the per-tap kernel {rms:.2e}")
        assert rms <= 2e-3, "halo-staged and per-tap launches differ by more than the summation order"
        o2 = torch.empty_like(o_h)
        hip.gemm(dev_bf(nhwc(x)), H.TiledW(dev_bf(w_h)), o2, epilogue=H.K_HALO32, **kw)
        assert torch.equal(o2, o_h), "halo-staged launch is not deterministic"
    # nearest x2 upsample + 3x3: the staged image is the low-resolution input
    for (n, c, cout, h, w) in [(2, 64, 320, 4, 32), (3, 128, 640, 6, 64), (5, 192, 320, 18, 32), (28, 640, 640, 18, 32)]:
        x = bf(rnd(n, c, h, w, seed=1))
        wt = bf(rnd(cout, c, 3, 3, seed=4) / math.sqrt(9 * c))
        b = rnd(cout, seed=6)
        M = n * 4 * h * w
        assert H.Hip.conv_halo_ok(M, cout, c, 2 * h, 2 * w, up=True)
        w_tap = wt.permute(0, 2, 3, 1).reshape(cout, -1)
        kw = dict(M=M, N=cout, K=9 * c, C1=c, mode=H.A_CONV3X3_UP, H=2 * h, Wd=2 * w, bias=dev_f(b))
        ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), wt, b, padding=1)
        o_h = torch.empty(M, cout, dtype=torch.bfloat16, device=DEV)
        o_t = torch.empty_like(o_h)
        hip.gemm(dev_bf(nhwc(x)), H.TiledW(dev_bf(conv_k_halo32(w_tap))), o_h, epilogue=H.K_HALO32, **kw)
        hip.gemm(dev_bf(nhwc(x)), H.TiledW(dev_bf(conv_k_cmajor(w_tap, 9))), o_t, epilogue=H.K_CMAJOR, **kw)
        check(from_nhwc(o_h, n, 2 * h, 2 * w), ref, what=f"halo up + conv3x3 {c}->{cout} {n}x{h}x{w}")
        d = (o_h.float() - o_t.float()).cpu()
        rms = float(d.pow(2).mean().sqrt() / o_t.float().pow(2).mean().sqrt().cpu())
        print(f"[halo] up + conv3x3 {c}->{cout} {n}x{h}x{w}: rms vs the per-tap kernel {rms:.2e}")
        assert rms <= 2e-3
    # outside the geometry: refused (W's K order belongs to this kernel)
    n, c, cout, h, w = 2, 64, 320, 16, 16
    x = bf(rnd(n, c, h, w, seed=1))
    wt = bf(rnd(cout, c, 3, 3, seed=4))
    out = torch.empty(n * h * w, cout, dtype=torch.bfloat16, device=DEV)
    assert not H.Hip.conv_halo_ok(n * h * w, cout, c, h, w)
    with pytest.raises(RuntimeError, match="HALO32"):
        hip.gemm(dev_bf(nhwc(x)), H.TiledW(dev_bf(conv_k_halo32(wt.permute(0, 2, 3, 1).reshape(cout, -1)))), out, M=n * h * w,
                 N=cout, K=9 * c, C1=c, mode=H.A_CONV3X3, H=h, Wd=w, epilogue=H.K_HALO32)


def test_box_calibration_burst_and_sustained(hip):
    """`Hip.calibrate_box` (bench.py's `box` object): the burst figure of the matrix pipe (wiw_calib_mfma, near-constant register
    operands), what it SUSTAINS on random operands under the board's power management (wiw_calib_mfma_random, ABI 17) and the
    device copy rate are plausible for an MI355X: sustained <= burst <= the 2.5 PFLOP/s dense figure (+ a margin for fp16's
    2.4 % higher burst), copy between 2 and 8 TB/s; bad arguments are refused."""
    r = hip.calibrate_box(sustain_s=0.3)
    assert r["cus"] >= 200
    assert 1000.0 < r["mfma_sustained_tflops"] <= r["mfma_tflops"] * 1.02 and r["mfma_tflops"] < 2600.0
    assert 2000.0 < r["copy_GBps"] < 8000.0
    assert hip.calibrate_box(sustain_s=0.0)["mfma_sustained_tflops"] is None
    out = torch.zeros(4, dtype=torch.float32, device=DEV)
    assert hip.lib.wiw_calib_mfma_random(None, 256, 1000, None, out.data_ptr()) == -1
    assert b"calib_mfma_random" in hip.lib.wiw_last_error()
