"""CLIP image encoder on the HIP kernels against the third-party module the reference calls
(`transformers.CLIPVisionModelWithProjection`, pipeline:213) run in fp32 on the CPU, and the preprocessing kernel against
the product's torch restatement of `_resize_with_antialiasing` — itself pinned to the reference by
tests/test_frontend.py (golden frontend_tiny.npz).  GPU only.

Tolerances: attention kernel as the other attention kernels (max 2e-2, rms 8e-3 of the reference on bf16-rounded
inputs); preprocessing: the operand rows are bf16 (2^-9) of an fp32 result that differs from the reference order of
operations by fp32 rounding only -> max 1.2e-2 / rms 4e-3; whole encoder (bf16 activations through L layers vs fp32):
rms <= 2e-2 of the embedding."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def bf(x):
    return x.to(torch.bfloat16).float()


def rel(out, ref):
    out, ref = out.detach().float().cpu(), ref.detach().float().cpu()
    assert out.shape == ref.shape and torch.isfinite(out).all()
    e = (out - ref).abs()
    return float(e.max() / (ref.abs().max() + 1e-30)), float(e.pow(2).mean().sqrt() / (ref.pow(2).mean().sqrt() + 1e-30))


@pytest.fixture(scope="module")
def hip():
    import wiw_amd  # noqa: F401
    from wiw_amd.hip import Hip

    return Hip(torch.device(DEV))


@pytest.mark.parametrize("seqs,S,heads,D", [(2, 257, 16, 80), (3, 50, 2, 64), (1, 17, 4, 16), (2, 64, 1, 128), (1, 5, 3, 48)])
def test_attn_small(hip, seqs, S, heads, D):
    C = heads * D
    Sp = -(-S // 16) * 16
    q, k, v = (bf(rnd(seqs, Sp, C, seed=s, scale=1.2)) for s in (1, 2, 3))
    qk = torch.cat([q, k], -1).reshape(seqs * Sp, 2 * C).to(DEV, torch.bfloat16).contiguous()
    vt = v.reshape(seqs * Sp, C).t().to(DEV, torch.bfloat16).contiguous()
    o = torch.full((seqs * Sp, C), float("nan"), dtype=torch.bfloat16, device=DEV)
    hip.attn_small(qk, 2 * C, C, vt, seqs * Sp, o, C, seqs, S, Sp, heads, D, D ** -0.5)

    def sp(t):
        return t[:, :S].reshape(seqs, S, heads, D).transpose(1, 2)

    ref = F.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(seqs, S, C)
    out = o.reshape(seqs, Sp, C)
    mx, rms = rel(out[:, :S], ref)
    print(f"[parity] attn_small seqs={seqs} S={S} heads={heads} d={D}: max_rel={mx:.3e} rms_rel={rms:.3e}")
    assert mx <= 2e-2 and rms <= 8e-3
    assert Sp == S or float(out[:, S:].float().abs().max()) == 0.0     # padding query rows are written as zeros


@pytest.mark.parametrize("B,H0,W0,P", [(2, 576, 1024, 14), (1, 96, 200, 32), (1, 224, 224, 14), (1, 150, 130, 14)])
def test_clip_preprocess_kernel(hip, B, H0, W0, P):
    from wiw_amd import frontend as FE
    from wiw_amd.clip import gaussian_taps

    x = torch.rand(B, 3, H0, W0, generator=torch.Generator().manual_seed(1)) * 2 - 1
    grid = 224 // P
    Sp = -(-(1 + grid * grid) // 16) * 16
    Kp = -(-(3 * P * P) // 64) * 64
    A = torch.zeros(B * Sp, Kp, dtype=torch.bfloat16, device=DEV)
    tmp = torch.empty(B * 3 * H0 * 224, dtype=torch.float32, device=DEV)
    hip.clip_preprocess(x.to(DEV), B, H0, W0, 224, P, gaussian_taps(W0, 224), gaussian_taps(H0, 224), FE.CLIP_MEAN,
                        [1.0 / s for s in FE.CLIP_STD], tmp, A, Sp, Kp)
    ref = FE.clip_preprocess(x)                                         # (B,3,224,224) fp32, the reference's order of ops
    cols = F.unfold(ref, kernel_size=P, stride=P).transpose(1, 2)       # (B, patches, 3*P*P), channel-major like the conv weight
    got = A.float().cpu().reshape(B, Sp, Kp)
    mx, rms = rel(got[:, 1:1 + grid * grid, :3 * P * P], cols)
    print(f"[parity] clip_preprocess {H0}x{W0} -> 224, patch {P}: max_rel={mx:.3e} rms_rel={rms:.3e}")
    assert mx <= 1.2e-2 and rms <= 4e-3
    assert float(got[:, 0].abs().max()) == 0.0 and float(got[:, 1 + grid * grid:].abs().max()) == 0.0


@pytest.mark.parametrize("hidden,heads,layers,inter,patch,act", [(320, 4, 2, 640, 32, "gelu"), (128, 2, 1, 256, 32, "quick_gelu"),
                                                                 (1280, 16, 2, 5120, 14, "gelu")])
def test_clip_encoder_matches_transformers_fp32(hip, hidden, heads, layers, inter, patch, act):
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection

    from wiw_amd import frontend as FE
    from wiw_amd.clip import CLIPVisionHIP

    torch.manual_seed(0)
    model = CLIPVisionModelWithProjection(CLIPVisionConfig(hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers,
                                                           num_attention_heads=heads, image_size=224, patch_size=patch,
                                                           projection_dim=1024, hidden_act=act)).eval()
    clip = CLIPVisionHIP.from_transformers(model, DEV, hip=hip)
    x = torch.rand(2, 3, 300, 420, generator=torch.Generator().manual_seed(2)) * 2 - 1
    out = clip(x)
    with torch.no_grad():
        ref = model(FE.clip_preprocess(x)).image_embeds
    mx, rms = rel(out, ref)
    print(f"[parity] CLIP encoder hidden={hidden} heads={heads} (d={hidden // heads}) L={layers} {act}: max_rel={mx:.3e} rms_rel={rms:.3e}")
    assert out.shape == (2, 1024) and rms <= 2e-2 and mx <= 4e-2
    # an image's embedding does not depend on what else is in the batch — bit for bit
    assert torch.equal(clip(x[:1]), out[:1])
