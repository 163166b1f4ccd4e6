"""Size-independent properties at the BASELINE config-A sizes (576x1024x14: latent 72x128, 28 frame-images), where the
CPU oracle is too slow to be the checker:
  * exact homogeneity of the bf16 GEMM / implicit-GEMM convolution: out(2 A) == 2 out(A) BIT FOR BIT (scaling by a
    power of two commutes with every rounding in the kernel) — exercises the 256x320 and 256x160 tiles, the LDS-DMA
    gather and the staged epilogue over all 258 048 rows;
  * candidate independence of the UNet forward (the B >= 2 contract): candidate 0 evaluated alone
    equals candidate 0 evaluated inside a batch of two — bit for bit (no kernel uses atomics; per-row arithmetic does not
    depend on the batch), and the same request gives the same bytes twice.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
FRAMES, H, W = 28, 72, 128
M = FRAMES * H * W            # 258 048


@pytest.fixture(scope="module")
def hip():
    import wiw_amd  # noqa: F401
    from wiw_amd.hip import Hip

    return Hip(torch.device(DEV))


def _rand(shape, seed, scale=1.0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(shape, generator=g, device=DEV) * scale).to(torch.bfloat16)


@pytest.mark.parametrize("mode,N,K,C", [("dense", 320, 1280, 1280), ("dense", 640, 320, 320), ("conv3x3", 320, 2880, 320),
                                        ("conv_t3", 320, 960, 320)])
def test_gemm_homogeneity_full_size(hip, mode, N, K, C):
    from wiw_amd import hip as Hh

    a = _rand((M, C), 1)
    w = _rand((N, K), 2, 1.0 / math.sqrt(K))
    r = _rand((M, N), 3)
    kw = dict(M=M, N=N, K=K, C1=C)
    if mode == "conv3x3":
        kw.update(mode=Hh.A_CONV3X3, H=H, Wd=W)
    elif mode == "conv_t3":
        kw.update(mode=Hh.A_CONV_T3, H=H, Wd=W, T=14)
    o1 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    o2 = torch.empty_like(o1)
    hip.gemm(a, w, o1, res1=r, ldr1=N, beta1=1.0, **kw)
    hip.gemm(a * 2, w, o2, res1=r * 2, ldr1=N, beta1=1.0, **kw)
    torch.cuda.synchronize()
    assert torch.isfinite(o1.float()).all() and float(o1.float().abs().mean()) > 0.1
    assert torch.equal(o2, o1 * 2), f"{mode} N={N} K={K}: out(2A) != 2 out(A)"


def test_unet_candidate_independence_full_size():
    import wiw_amd  # noqa: F401
    from wiw_amd.config import UNetConfig
    from wiw_amd.pipeline import SVDDenoiser
    from wiw_amd.unet import UNetHIP
    from wiw_amd.weights import random_state_dict_torch

    cfg = UNetConfig()
    unet = UNetHIP(cfg, random_state_dict_torch(cfg, 0, torch.device(DEV), torch.float32), DEV)
    den = SVDDenoiser(unet)
    g = torch.Generator().manual_seed(3)
    B, T, h, w = 2, cfg.num_frames, H, W
    il = torch.randn(B, 4, h, w, generator=g)
    ie = torch.randn(B, 1, cfg.cross_attention_dim, generator=g)
    nz = torch.randn(B, T, 4, h, w, generator=g)
    acts = np.array([[4] + [1, 2, 1, 3] * 3 + [1], [4] + [3] * 13])
    both = den.denoise(il, ie, nz, acts, num_steps=2).float().cpu()
    alone = den.denoise(il[:1], ie[:1], nz[:1], acts[:1], num_steps=2).float().cpu()
    again = den.denoise(il[:1], ie[:1], nz[:1], acts[:1], num_steps=2).float().cpu()
    assert torch.isfinite(both).all()

    def rel(a, b):
        return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())

    err, other = rel(both[0], alone[0]), rel(both[1], alone[0])
    print(f"[parity] full-size candidate independence: in-batch vs alone {err:.3e}, other candidate {other:.3e}")
    # every kernel is deterministic (no atomics) and evaluates a row with the same arithmetic whatever the batch:
    assert torch.equal(again, alone), "the same request must give the same bytes"
    assert torch.equal(both[0], alone[0]), "candidate 0 must not depend on what else is in the batch"
    assert other > 0.1
