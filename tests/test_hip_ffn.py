"""Parity of the fused FeedForward kernel `wiw_ffn_geglu_bf16` (csrc/ffn.hip) through the C ABI.  GPU only.

Oracle: FeedForward with GEGLU exactly as the reference evaluates it (dp/models/attention.py:1185-1243,
activations.py:93-123: `hidden, gate = proj(x).chunk(2); hidden * F.gelu(gate)`; then net.2), in fp32 on the SAME
16-bit-rounded operands, with the hidden activation rounded to the storage type where the un-fused path stores it.
Tolerance (the kernel gate of tests/test_hip_kernels.py): max|err| <= 1.2e-2 * max|ref|, rms <= 4e-3 * rms(ref) in bf16;
the fp16 build is held to 1.5e-3 / 6e-4.
Cases: single partial tile, exact tiles, more tiles than CUs (persistent blocks walk several tiles, H-waves run ahead across
tile boundaries), every epilogue form the UNet uses (plain residual + per-frame vector, AlphaBlender with two residuals),
rows_per_vec that is not a multiple of 16, the fused LayerNorm, strided input / output, bit-exact repeatability and
row independence (a row's result does not depend on M or on its position in a tile).
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
C, HID = 320, 1280


def _hip(dtype):
    import wiw_amd  # noqa: F401
    from wiw_amd.hip import Hip

    return Hip(torch.device(DEV), dtype)


@pytest.fixture(scope="module")
def hip():
    return _hip(torch.bfloat16)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def make_weights(dtype, seed=0, gamma=None, beta=None):
    """-> dict of reference tensors (rounded to `dtype`) and the packed device operands."""
    from wiw_amd.hip import FFN_CHUNK, TiledW
    from wiw_amd.unet import pack_geglu

    w0 = rnd(2 * HID, C, seed=seed + 1) / math.sqrt(C)
    b0 = rnd(2 * HID, seed=seed + 2) * 0.3
    w2 = (rnd(C, HID, seed=seed + 3) / math.sqrt(HID)).to(dtype).float()
    b2 = rnd(C, seed=seed + 4) * 0.3
    if gamma is not None:   # LayerNorm folded: gamma into W1, beta into b1 (what unet.py prepares)
        w0f, b0f = w0 * gamma[None, :], w0 @ beta + b0
    else:
        w0f, b0f = w0, b0
    w1p, b1p, _ = pack_geglu(w0f, b0f, FFN_CHUNK)
    w1p = w1p.to(dtype)
    ref_w0 = w0f.to(dtype).float()      # the rows the MFMAs multiply
    return dict(w0=ref_w0, b0=b0f, w2=w2, b2=b2,
                W1=TiledW(w1p.to(DEV).contiguous()), b1=b1p.to(DEV).contiguous(),
                W2=TiledW(w2.to(dtype).to(DEV).contiguous()), b2d=b2.to(DEV).contiguous())


def oracle(x, wt, dtype, *, ln=False, gamma=None, beta=None, rowvec=None, rows_per_vec=1, res1=None, beta1=0.0, res2=None,
           beta2=0.0, alpha=1.0, bias2=True):
    xf = x.float()
    if ln:   # the kernel normalises WITHOUT the affine (folded into W1 / b1) and rounds to the operand type
        xf = F.layer_norm(xf, (C,), None, None, 1e-5).to(dtype).float()
    proj = xf @ wt["w0"].t() + wt["b0"]
    hidden, gate = proj.chunk(2, dim=-1)
    h = (hidden * F.gelu(gate)).to(dtype).float()
    y = h @ wt["w2"].t()
    if bias2:
        y = y + wt["b2"]
    if rowvec is not None:
        y = y + rowvec[torch.arange(x.shape[0]) // rows_per_vec]
    y = alpha * y
    if res1 is not None:
        y = y + beta1 * res1.float()
    if res2 is not None:
        y = y + beta2 * res2.float()
    return y


def check(out, ref, dtype, what):
    out, ref = out.detach().float().cpu(), ref.detach().float().cpu()
    assert out.shape == ref.shape and torch.isfinite(out).all(), what
    err = (out - ref).abs()
    mx = float(err.max() / ref.abs().max())
    rms = float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    print(f"[parity] {what}: max_rel={mx:.3e} rms_rel={rms:.3e}")
    gate = (1.2e-2, 4e-3) if dtype == torch.bfloat16 else (1.5e-3, 6e-4)
    assert mx <= gate[0] and rms <= gate[1], f"{what}: max_rel={mx:.3e} rms_rel={rms:.3e}"


@pytest.mark.parametrize("M", [1, 100, 128, 1000, 4096, 256 * 128 + 77, 3 * 256 * 128 + 5])
def test_ffn_plain(hip, M):
    dt = torch.bfloat16
    wt = make_weights(dt)
    x = rnd(M, C, seed=10).to(dt)
    out = torch.full((M, C), float("nan"), dtype=dt, device=DEV)
    hip.ffn_geglu(x.to(DEV), wt["W1"], wt["b1"], wt["W2"], wt["b2d"], out, M)
    check(out, oracle(x, wt, dt), dt, f"ffn plain M={M}")


def test_ffn_identity_rows_catch_transposes(hip):
    """X = one-hot rows: out[m] = W2 . geglu(W1[:, m] + b1) + b2 — an asymmetric input that a swapped row / column mapping,
    a wrong value / gate pairing or a wrong hidden-chunk order cannot survive."""
    dt = torch.bfloat16
    wt = make_weights(dt, seed=50)
    M = C
    x = torch.eye(M, C).to(dt)
    out = torch.empty(M, C, dtype=dt, device=DEV)
    hip.ffn_geglu(x.to(DEV), wt["W1"], wt["b1"], wt["W2"], wt["b2d"], out, M)
    check(out, oracle(x, wt, dt), dt, "ffn one-hot rows")


@pytest.mark.parametrize("M,rpv", [(700, 100), (3 * 9216 // 8, 9216 // 8), (130, 7)])
def test_ffn_residual_and_frame_vector(hip, M, rpv):
    """The spatial FeedForward's epilogue: + hidden_states (res1) + per-frame vector (attention.py:565-582,
    transformer_temporal.py:352-353 folded in as DESIGN 3.1 describes)."""
    dt = torch.bfloat16
    wt = make_weights(dt, seed=20)
    x, r1 = rnd(M, C, seed=11).to(dt), rnd(M, C, seed=12).to(dt)
    nv = -(-M // rpv)
    rv = rnd(nv, C, seed=13)
    out = torch.empty(M, C, dtype=dt, device=DEV)
    hip.ffn_geglu(x.to(DEV), wt["W1"], wt["b1"], wt["W2"], wt["b2d"], out, M, rowvec=rv.to(DEV), rowvec_ld=C,
                  rows_per_vec=rpv, res1=r1.to(DEV), ldr1=C, beta1=1.0)
    check(out, oracle(x, wt, dt, rowvec=rv, rows_per_vec=rpv, res1=r1, beta1=1.0), dt, f"ffn res1+rowvec M={M} rpv={rpv}")


def test_ffn_alpha_blend_epilogue(hip):
    """The temporal FeedForward's epilogue = AlphaBlender (transformer_temporal.py:364-372):
    out = (1-a) * (ff + b2 + vec) + (1-a) * res1 + a * res2."""
    dt = torch.bfloat16
    wt = make_weights(dt, seed=30)
    M, rpv, am = 2000, 16 * 9, 0.37
    x, r1, r2 = rnd(M, C, seed=14).to(dt), rnd(M, C, seed=15).to(dt), rnd(M, C, seed=16).to(dt)
    rv = rnd(-(-M // rpv), C, seed=17)
    out = torch.empty(M, C, dtype=dt, device=DEV)
    hip.ffn_geglu(x.to(DEV), wt["W1"], wt["b1"], wt["W2"], wt["b2d"], out, M, rowvec=rv.to(DEV), rowvec_ld=C,
                  rows_per_vec=rpv, res1=r1.to(DEV), ldr1=C, beta1=1.0 - am, res2=r2.to(DEV), ldr2=C, beta2=am, alpha=1.0 - am)
    ref = oracle(x, wt, dt, rowvec=rv, rows_per_vec=rpv, res1=r1, beta1=1.0 - am, res2=r2, beta2=am, alpha=1.0 - am)
    check(out, ref, dt, "ffn AlphaBlender epilogue")


@pytest.mark.parametrize("M", [64, 1500, 2 * 256 * 128 + 300])
def test_ffn_fused_layernorm(hip, M):
    """ln=1: X is the RAW LayerNorm input (mean / std far from 0 / 1), gamma folded into W1 and beta into b1."""
    dt = torch.bfloat16
    gamma, beta = 1.0 + 0.2 * rnd(C, seed=40), 0.1 * rnd(C, seed=41)
    wt = make_weights(dt, seed=42, gamma=gamma, beta=beta)
    x = (rnd(M, C, seed=18) * 3.0 + 1.5 + rnd(M, 1, seed=19) * 4.0).to(dt)
    out = torch.empty(M, C, dtype=dt, device=DEV)
    hip.ffn_geglu(x.to(DEV), wt["W1"], wt["b1"], wt["W2"], wt["b2d"], out, M, ln=True, res1=x.to(DEV), ldr1=C, beta1=1.0)
    check(out, oracle(x, wt, dt, ln=True, res1=x, beta1=1.0), dt, f"ffn fused LayerNorm M={M}")
    # ... and against the reference's own order of operations: LayerNorm WITH its affine, rounded, then the un-folded weights
    wt_plain = make_weights(dt, seed=42)
    xn = F.layer_norm(x.float(), (C,), gamma, beta, 1e-5).to(dt)
    ref2 = oracle(xn, wt_plain, dt, res1=x, beta1=1.0)
    o, r = out.float().cpu(), ref2
    rms = float((o - r).pow(2).mean().sqrt() / r.pow(2).mean().sqrt())
    print(f"[parity] fused LayerNorm vs un-folded evaluation: rms_rel={rms:.3e}")
    assert rms <= 6e-3   # two 16-bit roundings in different places (folded gamma) — same error class as the kernel gate


def test_ffn_strided_and_no_bias(hip):
    dt = torch.bfloat16
    wt = make_weights(dt, seed=60)
    M, ldx, ldo = 777, 512, 384
    xs = rnd(M, ldx, seed=21).to(dt)
    outs = torch.zeros(M, ldo, dtype=dt, device=DEV)
    hip.ffn_geglu(xs.to(DEV), wt["W1"], wt["b1"], wt["W2"], None, outs, M, ldx=ldx, ldo=ldo)
    check(outs[:, :C], oracle(xs[:, :C], wt, dt, bias2=False), dt, "ffn strided, no b2")
    assert float(outs[:, C:].abs().max()) == 0.0   # columns past 320 untouched


def test_ffn_bit_exact_and_row_independent(hip):
    """Same bytes twice; and row m of a big launch == the same row evaluated alone in a small launch (the batch contract of
    DESIGN 5: every row's arithmetic is fixed, whatever the tile, block or launch size)."""
    dt = torch.bfloat16
    wt = make_weights(dt, seed=70)
    M = 300 * 128 + 19
    x = rnd(M, C, seed=22).to(dt).to(DEV)
    r1 = rnd(M, C, seed=23).to(dt).to(DEV)
    o1, o2 = torch.empty(M, C, dtype=dt, device=DEV), torch.empty(M, C, dtype=dt, device=DEV)
    for o in (o1, o2):
        hip.ffn_geglu(x, wt["W1"], wt["b1"], wt["W2"], wt["b2d"], o, M, res1=r1, ldr1=C, beta1=1.0, ln=False)
    assert torch.equal(o1, o2)
    lo, n = 128 * 257 + 5, 200   # rows from the middle of a later tile, evaluated as their own launch
    o3 = torch.empty(n, C, dtype=dt, device=DEV)
    hip.ffn_geglu(x[lo:lo + n].contiguous(), wt["W1"], wt["b1"], wt["W2"], wt["b2d"], o3, n, res1=r1[lo:lo + n].contiguous(),
                  ldr1=C, beta1=1.0)
    assert torch.equal(o1[lo:lo + n], o3)


def test_ffn_matches_unfused_gemm_chain(hip):
    """The fused kernel against the two-GEMM path it replaces (GEGLU epilogue + down projection), same operands."""
    from wiw_amd.hip import EPI_GEGLU, TiledW
    from wiw_amd.unet import pack_geglu

    dt = torch.bfloat16
    M = 5000
    w0 = rnd(2 * HID, C, seed=81) / math.sqrt(C)
    b0 = rnd(2 * HID, seed=82) * 0.3
    wt = make_weights(dt, seed=80)       # same seeds: w0 / b0 above equal make_weights' draws
    x, r1 = rnd(M, C, seed=24).to(dt).to(DEV), rnd(M, C, seed=25).to(dt).to(DEV)
    fused = torch.empty(M, C, dtype=dt, device=DEV)
    hip.ffn_geglu(x, wt["W1"], wt["b1"], wt["W2"], wt["b2d"], fused, M, res1=r1, ldr1=C, beta1=1.0)
    wp, bp, n_half = pack_geglu(w0, b0)
    g = torch.empty(M, HID, dtype=dt, device=DEV)
    hip.gemm(x, TiledW(wp.to(dt).to(DEV).contiguous()), g, M=M, N=wp.shape[0], K=C, C1=C, bias=bp.to(DEV), epilogue=EPI_GEGLU,
             n_out=HID)
    un = torch.empty(M, C, dtype=dt, device=DEV)
    hip.gemm(g, wt["W2"], un, M=M, N=C, K=HID, C1=HID, bias=wt["b2d"], res1=r1, ldr1=C, beta1=1.0)
    d = (fused.float() - un.float())
    rms = float(d.pow(2).mean().sqrt() / un.float().pow(2).mean().sqrt())
    print(f"[parity] fused vs un-fused chain: rms_rel={rms:.3e} max={float(d.abs().max()):.3e}")
    assert rms <= 2e-3     # both round the hidden activation and the output to bf16; only fp32 summation order differs


def test_ffn_fp16_build():
    dt = torch.float16
    hip16 = _hip(dt)
    wt = make_weights(dt, seed=90)
    M = 3000
    x, r1 = rnd(M, C, seed=26).to(dt), rnd(M, C, seed=27).to(dt)
    out = torch.empty(M, C, dtype=dt, device=DEV)
    hip16.ffn_geglu(x.to(DEV), wt["W1"], wt["b1"], wt["W2"], wt["b2d"], out, M, res1=r1.to(DEV), ldr1=C, beta1=1.0)
    check(out, oracle(x, wt, dt, res1=r1, beta1=1.0), dt, "ffn fp16 build")


@pytest.mark.parametrize("M,rpv", [(700, 100), (3 * 9216 // 8, 9216 // 8), (130, 7), (256 * 128 + 77, 4096)])
def test_ffn_fp32_residual_stream(hip, M, rpv):
    """`wiw_ffn_geglu_f32stream` (ABI 12): fp32 res1 / res2 and an fp32 output — fp32 accumulator + bias + vector + residuals,
    no rounding; a 16-bit output is the ONE rounding of the same sum; mixed 16-bit res1 + fp32 res2."""
    dt = torch.bfloat16
    wt = make_weights(dt, seed=21)
    x = rnd(M, C, seed=11).to(dt)
    r1, r2 = rnd(M, C, seed=12) * 3.0, rnd(M, C, seed=14)
    nv = -(-M // rpv)
    rv = rnd(nv, C, seed=13)
    a = 0.3
    kw = dict(rowvec=rv.to(DEV), rowvec_ld=C, rows_per_vec=rpv, ldr1=C, beta1=1.0 - a, ldr2=C, beta2=a, alpha=1.0 - a)
    ref = oracle(x, wt, dt, rowvec=rv, rows_per_vec=rpv, res1=r1, beta1=1.0 - a, res2=r2, beta2=a, alpha=1.0 - a)
    out32 = torch.full((M, C), float("nan"), dtype=torch.float32, device=DEV)
    hip.ffn_geglu(x.to(DEV), wt["W1"], wt["b1"], wt["W2"], wt["b2d"], out32, M, res1=r1.to(DEV), res2=r2.to(DEV), **kw)
    check(out32, ref, dt, f"ffn fp32 stream M={M} rpv={rpv}")
    out16 = torch.empty(M, C, dtype=dt, device=DEV)
    hip.ffn_geglu(x.to(DEV), wt["W1"], wt["b1"], wt["W2"], wt["b2d"], out16, M, res1=r1.to(DEV), res2=r2.to(DEV), **kw)
    assert torch.equal(out16.cpu(), out32.cpu().to(dt)), "16-bit output must be the ONE rounding of the fp32 result"
    r1h = r1.to(dt)
    hip.ffn_geglu(x.to(DEV), wt["W1"], wt["b1"], wt["W2"], wt["b2d"], out32, M, res1=r1h.to(DEV), res2=r2.to(DEV), **kw)
    check(out32, ref - (1.0 - a) * r1 + (1.0 - a) * r1h.float(), dt, f"ffn fp32 stream, 16-bit res1 M={M}")
    # against the staged 16-bit kernel on 16-bit residuals: the same value up to the staging's extra rounding
    r2h = r2.to(dt)
    o_st = torch.empty(M, C, dtype=dt, device=DEV)
    hip.ffn_geglu(x.to(DEV), wt["W1"], wt["b1"], wt["W2"], wt["b2d"], o_st, M, res1=r1h.to(DEV), res2=r2h.to(DEV), **kw)
    hip.ffn_geglu(x.to(DEV), wt["W1"], wt["b1"], wt["W2"], wt["b2d"], out32, M, res1=r1h.to(DEV), res2=r2h.float().to(DEV), **kw)
    d = (o_st.float() - out32).abs().max() / out32.abs().max()
    assert float(d) <= 1.2e-2


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [64, 1500, 9 * 128 * 20 + 77, 2 * 256 * 128 + 300])
def test_ffn_layernorm_reads_the_fp32_stream(dt, M):
    """`wiw_ffn_geglu_f32stream2` (ABI 16), f32 bit 3: X is the fp32 residual stream, the fused LayerNorm takes its moments
    from the fp32 row and packs the normalised 16-bit operand — bit-identical to the same kernel fed the f32in LayerNorm
    pass's job done in torch (normalise in fp32, round once), to which the standalone pass + 16-bit kernel also agree; `out16`
    is the rounding of `out`.  M values: one tile, several tiles per block, the last tile ragged."""
    hip = _hip(dt)
    gamma, beta = 1.0 + 0.2 * rnd(C, seed=40), 0.1 * rnd(C, seed=41)
    wt = make_weights(dt, seed=42, gamma=gamma, beta=beta)
    x = rnd(M, C, seed=18) * 3.0 + 1.5 + rnd(M, 1, seed=19) * 4.0          # fp32 rows, mean / std far from 0 / 1
    a = 0.3
    hs = rnd(M, C, seed=20)
    out = torch.full((M, C), float("nan"), dtype=torch.float32, device=DEV)
    o16 = torch.full((M, C), float("nan"), dtype=dt, device=DEV)
    kw = dict(res1=x.to(DEV), ldr1=C, beta1=1.0 - a, res2=hs.to(DEV), ldr2=C, beta2=a, alpha=1.0 - a)
    hip.ffn_geglu(x.to(DEV), wt["W1"], wt["b1"], wt["W2"], wt["b2d"], out, M, ln=True, out16=o16, **kw)
    ref = oracle(x, wt, dt, ln=True, res1=x, beta1=1.0 - a, res2=hs, beta2=a, alpha=1.0 - a)
    check(out, ref, dt, f"ffn LayerNorm on the fp32 stream M={M} {dt}")
    assert torch.equal(o16.cpu(), out.cpu().to(dt)), "out16 must be the rounding of out"
    # the normalised operand: the kernel on the PRE-normalised, rounded rows (ln = 0, same folded weights) gives the same bits
    # wherever torch's layer_norm and the kernel's two-pass moments round alike; everywhere within an ulp of the operand
    xn = F.layer_norm(x, (C,), None, None, 1e-5).to(dt)
    out_b = torch.empty(M, C, dtype=torch.float32, device=DEV)
    hip.ffn_geglu(xn.to(DEV), wt["W1"], wt["b1"], wt["W2"], wt["b2d"], out_b, M, **kw)
    d = float((out - out_b).abs().max() / out_b.abs().max())
    print(f"[parity] in-kernel fp32 LayerNorm vs torch-normalised operand: max_rel={d:.2e}")
    assert d <= (4e-3 if dt == torch.bfloat16 else 5e-4)
    # 16-bit output + twice: bit-repeatable
    o2 = torch.empty(M, C, dtype=dt, device=DEV)
    hip.ffn_geglu(x.to(DEV), wt["W1"], wt["b1"], wt["W2"], wt["b2d"], o2, M, ln=True, **kw)
    assert torch.equal(o2.cpu(), out.cpu().to(dt))
    out2 = torch.empty_like(out)
    hip.ffn_geglu(x.to(DEV), wt["W1"], wt["b1"], wt["W2"], wt["b2d"], out2, M, ln=True, **kw)
    assert torch.equal(out2, out)


def test_ffn_out16_without_fp32_x(hip):
    """out16 beside a 16-bit X (the ff_in of the fp32-stream mode behind a separate LayerNorm pass)."""
    dt = torch.bfloat16
    M = 3000
    wt = make_weights(dt, seed=21)
    x = rnd(M, C, seed=11).to(dt)
    r1 = rnd(M, C, seed=12) * 3.0
    out = torch.empty(M, C, dtype=torch.float32, device=DEV)
    o16 = torch.empty(M, C, dtype=dt, device=DEV)
    hip.ffn_geglu(x.to(DEV), wt["W1"], wt["b1"], wt["W2"], wt["b2d"], out, M, res1=r1.to(DEV), ldr1=C, beta1=1.0, out16=o16)
    check(out, oracle(x, wt, dt, res1=r1, beta1=1.0), dt, "ffn fp32 out + out16")
    assert torch.equal(o16.cpu(), out.cpu().to(dt))


# ----------------------------------------------------------------------------------------------------------------------
# wiw_ffn32_geglu (csrc/ffn32.hip, ABI 14): the same operator on 32x32x16 MFMAs, everything in registers.  Same oracle, same
# gates.  (Opt-in in the UNet — WIW_FFN32=1 — because it is not faster than ffn.hip; it is kept correct.)
# ----------------------------------------------------------------------------------------------------------------------
def make_weights32(dtype, seed=0, gamma=None, beta=None):
    from wiw_amd.unet import pack_ffn32

    w0 = rnd(2 * HID, C, seed=seed + 1) / math.sqrt(C)
    b0 = rnd(2 * HID, seed=seed + 2) * 0.3
    w2 = rnd(C, HID, seed=seed + 3) / math.sqrt(HID)
    b2 = rnd(C, seed=seed + 4) * 0.3
    if gamma is not None:
        w0f, b0f = w0 * gamma[None, :], w0 @ beta + b0
    else:
        w0f, b0f = w0, b0
    W1s, b1s, W2s = pack_ffn32(w0f.to(DEV), b0f.to(DEV), w2.to(DEV), dtype)
    return dict(w0=w0f.to(dtype).float(), b0=b0f, w2=w2.to(dtype).float(), b2=b2, W1=W1s, b1=b1s, W2=W2s, b2d=b2.to(DEV).contiguous())


def test_sw16_tiling_round_trips():
    from wiw_amd.hip import TiledW

    w = rnd(64 * 3, 320, seed=3).to(torch.bfloat16).to(DEV)
    for sw16 in (False, True):
        assert torch.equal(TiledW(w, sw16=sw16).untiled(), w)
    assert not torch.equal(TiledW(w, sw16=True).data, TiledW(w).data)


@pytest.mark.parametrize("M", [1, 100, 128, 1000, 256 * 128 + 77, 3 * 256 * 128 + 5])
def test_ffn32_plain(hip, M):
    dt = torch.bfloat16
    wt = make_weights32(dt)
    x = rnd(M, C, seed=10).to(dt)
    out = torch.full((M, C), float("nan"), dtype=dt, device=DEV)
    hip.ffn32_geglu(x.to(DEV), wt["W1"], wt["b1"], wt["W2"], wt["b2d"], out, M)
    check(out, oracle(x, wt, dt), dt, f"ffn32 plain M={M}")


def test_ffn32_identity_rows_catch_transposes(hip):
    dt = torch.bfloat16
    wt = make_weights32(dt, seed=50)
    x = torch.eye(C, C).to(dt)
    out = torch.empty(C, C, dtype=dt, device=DEV)
    hip.ffn32_geglu(x.to(DEV), wt["W1"], wt["b1"], wt["W2"], wt["b2d"], out, C)
    check(out, oracle(x, wt, dt), dt, "ffn32 one-hot rows")


@pytest.mark.parametrize("M,rpv", [(700, 100), (3 * 9216 // 8, 9216 // 8), (130, 7)])
def test_ffn32_epilogues(hip, M, rpv):
    """Residual + per-frame vector; AlphaBlender with two residuals; strided, no b2."""
    dt = torch.bfloat16
    wt = make_weights32(dt, seed=20)
    x, r1, r2 = rnd(M, C, seed=11).to(dt), rnd(M, C, seed=12).to(dt), rnd(M, C, seed=16).to(dt)
    rv = rnd(-(-M // rpv), C, seed=13)
    out = torch.empty(M, C, dtype=dt, device=DEV)
    hip.ffn32_geglu(x.to(DEV), wt["W1"], wt["b1"], wt["W2"], wt["b2d"], out, M, rowvec=rv.to(DEV), rowvec_ld=C, rows_per_vec=rpv,
                    res1=r1.to(DEV), ldr1=C, beta1=1.0)
    check(out, oracle(x, wt, dt, rowvec=rv, rows_per_vec=rpv, res1=r1, beta1=1.0), dt, f"ffn32 res1+rowvec M={M} rpv={rpv}")
    am = 0.37
    hip.ffn32_geglu(x.to(DEV), wt["W1"], wt["b1"], wt["W2"], wt["b2d"], out, M, rowvec=rv.to(DEV), rowvec_ld=C, rows_per_vec=rpv,
                    res1=r1.to(DEV), ldr1=C, beta1=1.0 - am, res2=r2.to(DEV), ldr2=C, beta2=am, alpha=1.0 - am)
    check(out, oracle(x, wt, dt, rowvec=rv, rows_per_vec=rpv, res1=r1, beta1=1.0 - am, res2=r2, beta2=am, alpha=1.0 - am), dt,
          f"ffn32 AlphaBlender M={M}")
    ldx, ldo = 512, 384
    xs = rnd(M, ldx, seed=21).to(dt)
    outs = torch.zeros(M, ldo, dtype=dt, device=DEV)
    hip.ffn32_geglu(xs.to(DEV), wt["W1"], wt["b1"], wt["W2"], None, outs, M, ldx=ldx, ldo=ldo)
    check(outs[:, :C], oracle(xs[:, :C], wt, dt, bias2=False), dt, "ffn32 strided, no b2")
    assert float(outs[:, C:].abs().max()) == 0.0


@pytest.mark.parametrize("M", [64, 1500, 2 * 256 * 128 + 300])
def test_ffn32_fused_layernorm(hip, M):
    dt = torch.bfloat16
    gamma, beta = 1.0 + 0.2 * rnd(C, seed=40), 0.1 * rnd(C, seed=41)
    wt = make_weights32(dt, seed=42, gamma=gamma, beta=beta)
    x = (rnd(M, C, seed=18) * 3.0 + 1.5 + rnd(M, 1, seed=19) * 4.0).to(dt)
    out = torch.empty(M, C, dtype=dt, device=DEV)
    hip.ffn32_geglu(x.to(DEV), wt["W1"], wt["b1"], wt["W2"], wt["b2d"], out, M, ln=True, res1=x.to(DEV), ldr1=C, beta1=1.0)
    check(out, oracle(x, wt, dt, ln=True, res1=x, beta1=1.0), dt, f"ffn32 fused LayerNorm M={M}")


def test_ffn32_bit_exact_row_independent_and_fp32_stream(hip):
    dt = torch.bfloat16
    wt = make_weights32(dt, seed=70)
    M = 300 * 128 + 19
    x = rnd(M, C, seed=22).to(dt).to(DEV)
    r1 = rnd(M, C, seed=23).to(dt).to(DEV)
    o1, o2 = torch.empty(M, C, dtype=dt, device=DEV), torch.empty(M, C, dtype=dt, device=DEV)
    for o in (o1, o2):
        hip.ffn32_geglu(x, wt["W1"], wt["b1"], wt["W2"], wt["b2d"], o, M, res1=r1, ldr1=C, beta1=1.0)
    assert torch.equal(o1, o2)
    lo, n = 128 * 257 + 5, 200
    o3 = torch.empty(n, C, dtype=dt, device=DEV)
    hip.ffn32_geglu(x[lo:lo + n].contiguous(), wt["W1"], wt["b1"], wt["W2"], wt["b2d"], o3, n, res1=r1[lo:lo + n].contiguous(), ldr1=C, beta1=1.0)
    assert torch.equal(o1[lo:lo + n], o3)
    # fp32 residual stream: fp32 res1 and output; the 16-bit output of the same call is the ONE rounding of the fp32 result
    r32 = rnd(M, C, seed=24) * 3.0
    o32 = torch.full((M, C), float("nan"), dtype=torch.float32, device=DEV)
    hip.ffn32_geglu(x, wt["W1"], wt["b1"], wt["W2"], wt["b2d"], o32, M, res1=r32.to(DEV), ldr1=C, beta1=1.0)
    check(o32, oracle(x.cpu(), wt, dt, res1=r32, beta1=1.0), dt, "ffn32 fp32 stream")
    o16 = torch.empty(M, C, dtype=dt, device=DEV)
    hip.ffn32_geglu(x, wt["W1"], wt["b1"], wt["W2"], wt["b2d"], o16, M, res1=r32.to(DEV), ldr1=C, beta1=1.0)
    assert torch.equal(o16.cpu(), o32.cpu().to(dt))


def test_ffn32_fp16_build():
    dt = torch.float16
    hip16 = _hip(dt)
    wt = make_weights32(dt, seed=90)
    M = 3000
    x, r1 = rnd(M, C, seed=26).to(dt), rnd(M, C, seed=27).to(dt)
    out = torch.empty(M, C, dtype=dt, device=DEV)
    hip16.ffn32_geglu(x.to(DEV), wt["W1"], wt["b1"], wt["W2"], wt["b2d"], out, M, res1=r1.to(DEV), ldr1=C, beta1=1.0)
    check(out, oracle(x, wt, dt, res1=r1, beta1=1.0), dt, "ffn32 fp16 build")
