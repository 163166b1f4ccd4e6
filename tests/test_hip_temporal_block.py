"""The fused temporal block (temporal.hip: LayerNorm folded + per-head Q/K/V projection + 14x14 attention, one kernel)
against a plain fp32 restatement of `TemporalBasicTransformerBlock.norm1 + attn1` (reference attention.py:735-737,
attention_processor.py:2358-2391) on the same bf16-rounded inputs.

Tolerances (as for the other attention kernels in tests/test_hip_kernels.py): max <= 2e-2 * max|ref|, rms <= 8e-3 —
Q, K, V and the probabilities are rounded to bf16 between the MFMA stages, as they are when the unfused chain stores
them to HBM.  The CPU test pins the LayerNorm fold algebra and the per-head packing (no GPU needed)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

DEV = "cuda:0"


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def bf(x):
    return x.to(torch.bfloat16).float()


def reference(x, gamma, beta, wq, wk, wv, heads, eps=1e-5):
    """x (B,T,S,C) -> attention output (B,T,S,C), per site over the T frames."""
    B, T, S, C = x.shape
    a = F.layer_norm(x, (C,), gamma, beta, eps)
    q, k, v = a @ wq.t(), a @ wk.t(), a @ wv.t()

    def perm(t):  # (B,T,S,C) -> (B*S, heads, T, 64)
        return t.permute(0, 2, 1, 3).reshape(B * S, T, heads, 64).transpose(1, 2)

    o = F.scaled_dot_product_attention(perm(q), perm(k), perm(v))
    return o.transpose(1, 2).reshape(B, S, T, C).permute(0, 2, 1, 3)


def test_pack_temporal_qkv_fold_algebra():
    """LayerNorm(x) . W_n == rstd * (x . W'_n - mean * s_n) + t_n with the packed operands (fp64 check)."""
    import wiw_amd  # noqa: F401
    from wiw_amd.unet import pack_temporal_qkv

    C, heads = 128, 2
    wq, wk, wv = (rnd(C, C, seed=s) / math.sqrt(C) for s in (1, 2, 3))
    gamma, beta = 1 + 0.3 * rnd(C, seed=4), 0.2 * rnd(C, seed=5)
    from wiw_amd.hip import TiledW, tile_weight, untile_weight

    wg, fold = pack_temporal_qkv(wq, wk, wv, gamma, beta, tiled=False)
    assert wg.shape == (heads * 192, C) and wg.dtype == torch.bfloat16 and fold.shape == (heads, 512)
    # the kernel's operand = the same matrix in the tiled LDS-DMA layout (1-KiB blocks, swizzled chunks): a bijection
    tiled, _ = pack_temporal_qkv(wq, wk, wv, gamma, beta)
    assert tiled.shape == (heads * 192 * C,) and torch.equal(untile_weight(tiled, heads * 192, C), wg)
    odd = torch.randn(13, 128).to(torch.bfloat16)              # N not a multiple of 8: zero rows pad the last block
    assert torch.equal(TiledW(odd).untiled(), odd) and tile_weight(odd).numel() == 16 * 128
    blk = tile_weight(wg).reshape(-1, 8, 8, 8)                 # [block][row r][position p][8 elements]
    assert torch.equal(blk[0, 3, 5], wg[3, (5 ^ 3) * 8:(5 ^ 3) * 8 + 8])   # position p of row r holds chunk p ^ r
    x = rnd(7, C, seed=6) * 2 + 0.5
    mean, var = x.double().mean(-1, keepdim=True), x.double().var(-1, unbiased=False, keepdim=True)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    s, t = fold[:, :192].reshape(-1).double(), fold[:, 192:384].reshape(-1).double()
    folded = rstd * (x.double() @ wg.double().t() - mean * s) + t
    a = F.layer_norm(x.double(), (C,), gamma.double(), beta.double(), 1e-5)
    for h in range(heads):
        for j, wm in enumerate((wq, wk, wv)):
            ref = a @ wm.double()[h * 64:(h + 1) * 64].t()
            got = folded[:, h * 192 + j * 64: h * 192 + (j + 1) * 64]
            # the only difference is the bf16 rounding of W * gamma (2^-9 relative per weight)
            assert float((got - ref).abs().max()) <= 2e-2 * float(ref.abs().max())
    assert float(fold[:, 384:].abs().max()) == 0.0
    # the fp16 build packs the same operands in its own 16-bit type (libwiwsvd_f16.so); the tiling is a pure
    # permutation of 16-bit words, so it is dtype-blind
    wg16, fold16 = pack_temporal_qkv(wq, wk, wv, gamma, beta, tiled=False, dtype=torch.float16)
    assert wg16.dtype == torch.float16 and torch.equal(fold16[:, 192:384], fold[:, 192:384])
    assert float((wg16.float() - wg.float()).abs().max()) <= 2.0 ** -8 * float(wg.float().abs().max())
    assert torch.equal(fold16[:, :192].reshape(-1), wg16.float().sum(dim=1))      # s is the sum of the ROUNDED rows
    assert torch.equal(TiledW(wg16).untiled(), wg16)


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,S,heads", [(2, 4, 32, 1), (1, 14, 80, 5), (1, 14, 16, 10), (1, 14, 16, 20), (2, 3, 8, 2),
                                         (1, 8, 40, 2), (3, 14, 8, 1), (1, 1, 16, 1), (2, 14, 400, 5)])
def test_temporal_attn_block(B, T, S, heads):
    import wiw_amd  # noqa: F401
    from wiw_amd.hip import Hip
    from wiw_amd.unet import pack_temporal_qkv

    hip = Hip(torch.device(DEV))
    C = heads * 64
    x = bf(rnd(B, T, S, C, seed=1) * 1.5 + 0.3)
    x[..., 5] += 4.0                                            # a channel with an offset: the fold must remove the mean
    wq, wk, wv = (rnd(C, C, seed=s) / math.sqrt(C) for s in (2, 3, 4))
    gamma, beta = 1 + 0.2 * rnd(C, seed=5), 0.2 * rnd(C, seed=6)
    wg, fold = pack_temporal_qkv(wq, wk, wv, gamma, beta)
    o = torch.full((B * T * S, C), float("nan"), dtype=torch.bfloat16, device=DEV)
    hip.temporal_attn_block(x.reshape(-1, C).to(DEV, torch.bfloat16).contiguous(), wg.to(DEV), fold.to(DEV), o, C, B, T, S,
                            heads, 1e-5, 0.125)
    torch.cuda.synchronize()
    # reference on what the kernel multiplies: the gamma-folded, bf16-rounded weights (un-folded by gamma in fp32)
    ref = reference(x, gamma, beta, bf(wq), bf(wk), bf(wv), heads)
    out = o.float().cpu().reshape(B, T, S, C)
    assert torch.isfinite(out).all(), "rows left unwritten or non-finite"
    err = (out - ref).abs()
    mx = float(err.max() / ref.abs().max())
    rms = float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    print(f"[parity] temporal_attn_block B={B} T={T} S={S} heads={heads}: max_rel={mx:.3e} rms_rel={rms:.3e}")
    assert mx <= 2e-2 and rms <= 8e-3


@pytest.mark.gpu
def test_temporal_attn_block_matches_unfused_chain_north_star_shape():
    """18 432 sites x 14 frames x 320 channels: the fused kernel against the unfused HIP chain (LayerNorm kernel -> QKV
    GEMM -> wiw_attn_temporal_bf16) and against the fp32 restatement on a strided site sample."""
    import wiw_amd  # noqa: F401
    from wiw_amd.hip import Hip
    from wiw_amd.unet import pack_temporal_qkv

    hip = Hip(torch.device(DEV))
    B, T, S, heads = 2, 14, 9216, 5
    C = heads * 64
    g = torch.Generator(device=DEV).manual_seed(1)
    x = (torch.randn(B * T * S, C, generator=g, device=DEV) * 1.5 + 0.3).to(torch.bfloat16)
    wq, wk, wv = (rnd(C, C, seed=s) / math.sqrt(C) for s in (2, 3, 4))
    gamma, beta = 1 + 0.2 * rnd(C, seed=5), 0.2 * rnd(C, seed=6)
    wg, fold = pack_temporal_qkv(wq, wk, wv, gamma, beta)
    o = torch.empty(B * T * S, C, dtype=torch.bfloat16, device=DEV)
    hip.temporal_attn_block(x, wg.to(DEV), fold.to(DEV), o, C, B, T, S, heads, 1e-5, 0.125)
    # unfused chain on the device
    a = hip.layernorm(x, B * T * S, C, gamma.to(DEV), beta.to(DEV))
    wqkv = torch.cat([wq, wk, wv]).to(DEV, torch.bfloat16).contiguous()
    qkv = torch.empty(B * T * S, 3 * C, dtype=torch.bfloat16, device=DEV)
    hip.gemm(a, wqkv, qkv, M=B * T * S, N=3 * C, K=C, C1=C)
    o2 = torch.empty_like(o)
    hip.attn_temporal(qkv, 3 * C, o2, C, B, T, S, heads, 0.125)
    torch.cuda.synchronize()
    d = (o.float() - o2.float())
    rms_chain = float(d.pow(2).mean().sqrt() / o2.float().pow(2).mean().sqrt())
    sites = torch.arange(0, S, 37)
    xs = x.reshape(B, T, S, C)[:, :, sites.to(DEV)].float().cpu()
    ref = reference(xs, gamma, beta, bf(wq), bf(wk), bf(wv), heads)
    out = o.reshape(B, T, S, C)[:, :, sites.to(DEV)].float().cpu()
    err = (out - ref).abs()
    mx = float(err.max() / ref.abs().max())
    rms = float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    print(f"[parity] temporal_attn_block 18432x14x320: vs fp32 max_rel={mx:.3e} rms_rel={rms:.3e}; vs unfused HIP chain rms={rms_chain:.3e}")
    assert torch.isfinite(o.float()).all()
    assert mx <= 2e-2 and rms <= 8e-3 and rms_chain <= 1.2e-2


@pytest.mark.gpu
def test_ring_form_at_every_width_in_a_subprocess():
    """C <= 320 takes the weight-resident kernel by default; the ring kernel serves the same shapes under WIW_TEMPORAL_RING=1
    (the A/B knob).  The library reads the knob once per process, so the parity cases above are re-run in a child process
    with it set: both forms stay pinned to the same reference."""
    import os
    import subprocess
    import sys

    if os.environ.get("WIW_TEMPORAL_RING"):
        pytest.skip("already the child process")
    env = dict(os.environ, WIW_TEMPORAL_RING="1")
    r = subprocess.run([sys.executable, "-m", "pytest", __file__, "-q", "-m", "gpu", "-k", "test_temporal_attn_block", "-x"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
