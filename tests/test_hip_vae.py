"""Temporal VAE on the HIP kernels against the fp32 functional restatement in
oracle/vae_oracle.py, which tests/test_frontend.py pins to the reference VAE through tests/golden/frontend_tiny.npz.

Tolerances: the HIP path keeps activations in bf16 between layers (as the reference does under its bf16
`weight_dtype`); against the fp32 chain the accumulated rounding of ~60 layers is gated at 3e-2 relative rms
for the decoded frames / encoded latents, 1.2e-2 max / 4e-3 rms for single kernels (same as test_hip_kernels).
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import vae_oracle as VO

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
VCFG = dict(block_out_channels=(64, 128, 128, 128), layers_per_block=2)


@pytest.fixture(scope="module")
def hip():
    import wiw_amd  # noqa: F401
    from wiw_amd.hip import Hip

    return Hip(torch.device(DEV))


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def rel(out, ref):
    out, ref = out.detach().float().cpu(), ref.detach().float().cpu()
    assert out.shape == ref.shape and torch.isfinite(out).all()
    err = (out - ref).abs()
    return float(err.max() / (ref.abs().max() + 1e-30)), float(err.pow(2).mean().sqrt() / (ref.pow(2).mean().sqrt() + 1e-30))


@pytest.mark.parametrize("rows,cols", [(7, 64), (33, 512), (5, 9216), (3, 16384), (4, 1028)])
def test_softmax_rows(hip, rows, cols):
    x = rnd(rows, cols, seed=1) * 4.0
    x[0, 3] = 30.0   # a dominant logit
    out = torch.empty(rows, cols, dtype=torch.bfloat16, device=DEV)
    hip.softmax_rows(x.to(DEV).contiguous(), cols, rows, cols, out, cols)
    mx, rms = rel(out, torch.softmax(x, dim=-1))
    print(f"[parity] softmax_rows {rows}x{cols}: max_rel={mx:.3e} rms_rel={rms:.3e}")
    assert mx <= 4e-3 and rms <= 4e-3   # bf16 rounding of the output only


@pytest.mark.parametrize("B,T,h,w", [(1, 4, 8, 16), (2, 3, 5, 7), (1, 1, 4, 4)])
def test_time_conv_out(hip, B, T, h, w):
    y = rnd(B * T, 3, h, w, seed=2)
    wt, b = rnd(3, 3, 3, 1, 1, seed=3) / 3.0, rnd(3, seed=4)
    yin = torch.full((B * T * h * w, 4), float("nan"))      # column 3 is padding the kernel must ignore
    yin[:, :3] = y.permute(0, 2, 3, 1).reshape(-1, 3)
    out = torch.empty(B * T, 3, h, w, dtype=torch.float32, device=DEV)
    hip.vae_time_conv_out(yin.to(DEV), 4, wt[:, :, :, 0, 0].contiguous().to(DEV), b.to(DEV), B * T, T, h * w, out)
    ref = F.conv3d(y.reshape(B, T, 3, h, w).permute(0, 2, 1, 3, 4), wt, b, padding=(1, 0, 0)).permute(0, 2, 1, 3, 4)
    mx, rms = rel(out, ref.reshape(B * T, 3, h, w))
    assert mx <= 1e-5 and rms <= 1e-5


def test_nchw_to_nhwc_pad(hip):
    x = rnd(3, 4, 6, 10, seed=5)
    out = torch.empty(3 * 60, 64, dtype=torch.bfloat16, device=DEV)
    hip.nchw_to_nhwc(x.to(DEV), 3, 4, 60, 0.5, 64, out)
    ref = torch.zeros(3 * 60, 64)
    ref[:, :4] = (0.5 * x).permute(0, 2, 3, 1).reshape(-1, 4)
    assert torch.equal(out.float().cpu(), ref.to(torch.bfloat16).float())


def test_conv_stride2_bottom_right_pad(hip):
    """Downsample2D(padding=0): F.pad(x, (0,1,0,1)) then a stride-2 3x3 convolution (downsampling.py:132-150)."""
    from wiw_amd import hip as H

    n, c, co, h, w = 2, 64, 128, 8, 16
    x = rnd(n, c, h, w, seed=1).to(torch.bfloat16).float()
    wt = (rnd(co, c, 3, 3, seed=2) / math.sqrt(9 * c)).to(torch.bfloat16).float()
    b = rnd(co, seed=3)
    Mo = n * (h // 2) * (w // 2)
    out = torch.empty(Mo, co, dtype=torch.bfloat16, device=DEV)
    hip.gemm(x.permute(0, 2, 3, 1).reshape(-1, c).to(DEV, torch.bfloat16).contiguous(),
             wt.permute(0, 2, 3, 1).reshape(co, -1).to(DEV, torch.bfloat16).contiguous(), out, M=Mo, N=co, K=9 * c, C1=c,
             mode=H.A_CONV3X3_S2P, H=h // 2, Wd=w // 2, bias=b.to(DEV))
    ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), wt, b, stride=2)
    mx, rms = rel(out.float().cpu().reshape(n, h // 2, w // 2, co).permute(0, 3, 1, 2), ref)
    print(f"[parity] conv3x3 stride 2, pad (0,1,0,1): max_rel={mx:.3e} rms_rel={rms:.3e}")
    assert mx <= 1.2e-2 and rms <= 4e-3


@pytest.fixture(scope="module")
def vae_pair(hip):
    from wiw_amd import frontend as FE
    from wiw_amd.vae import VAEHIP

    sd = FE.vae_random_state_dict(31, **VCFG)
    return VAEHIP(sd, DEV, hip=hip, **VCFG), {k: torch.from_numpy(v) for k, v in sd.items()}


def test_vae_decode_matches_fp32_chain(vae_pair):
    from wiw_amd import frontend as FE

    vae, sd = vae_pair
    T, h, w = 4, 16, 32
    z = rnd(T, 4, h, w, seed=7) * 3.0
    out = vae.decode_frames(z)
    ref = VO.vae_decode(sd, z, T, len(VCFG["block_out_channels"]), VCFG["layers_per_block"])
    mx, rms = rel(out, ref)
    print(f"[parity] VAE decode {T}x{8 * h}x{8 * w}: max_rel={mx:.3e} rms_rel={rms:.3e}")
    assert out.shape == (T, 3, 8 * h, 8 * w) and rms <= 3e-2
    # decode() = per-clip decode of latents / scaling_factor (pipeline:282-309)
    lat = torch.stack([z, z.flip(0)]) * vae.scaling_factor
    both = vae.decode(lat)
    assert both.shape == (2, T, 3, 8 * h, 8 * w)
    assert rel(both[0], out)[1] <= 1e-2      # (z * s) / s differs from z in the last fp32 bit -> bf16 rounding flips only


def test_vae_encode_matches_fp32_chain(vae_pair):
    from wiw_amd import frontend as FE

    vae, sd = vae_pair
    x = torch.tanh(rnd(2, 3, 128, 256, seed=9))
    out = vae.encode_mode(x)
    ref = VO.vae_encode_mode(sd, x, len(VCFG["block_out_channels"]), VCFG["layers_per_block"])
    mx, rms = rel(out, ref)
    print(f"[parity] VAE encode 2x128x256: max_rel={mx:.3e} rms_rel={rms:.3e}")
    assert out.shape == (2, 4, 16, 32) and rms <= 3e-2


def test_hip_frontend_protocol(vae_pair):
    """HIPFrontend satisfies server.worker.Frontend: numpy in / numpy out, shapes of pipeline:239, 282-309."""
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection

    from wiw_amd.vae import HIPFrontend

    vae, _ = vae_pair
    torch.manual_seed(0)
    clip = CLIPVisionModelWithProjection(CLIPVisionConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=1,
                                                          num_attention_heads=2, image_size=224, patch_size=32,
                                                          projection_dim=1024)).eval()
    fe = HIPFrontend(vae, clip, dtype=torch.float32, clip="torch")   # 32-wide toy CLIP: below the HIP kernels' 64-channel granularity
    rs = np.random.RandomState(0)
    x = np.tanh(rs.standard_normal((2, 3, 128, 256))).astype(np.float32)
    lat, emb = fe.encode(x, rs.standard_normal(x.shape).astype(np.float32), 0.02)
    assert lat.shape == (2, 4, 16, 32) and emb.shape == (2, 1, 1024) and lat.dtype == np.float32
    lat = rs.standard_normal((2, 4, 4, 16, 32)).astype(np.float32)
    fr = fe.decode(lat)
    assert fr.shape == (2, 4, 3, 128, 256) and np.isfinite(fr).all()
    # device-side quantisation == frames_to_pil on the host, on the SAME decoded frames (the kernels are deterministic —
    # no atomics — so a second decode would give the same bytes; the comparison below only allows for decode_uint8
    # running clip by clip)
    from wiw_amd.server import plumbing as P
    from wiw_amd.vae import frames_to_uint8_device
    frd = vae.decode(torch.from_numpy(lat))[0]
    ref = np.stack([np.asarray(im) for im in P.frames_to_pil(frd.cpu().numpy())])
    assert np.array_equal(frames_to_uint8_device(frd).cpu().numpy(), ref)
    u8 = fe.decode_uint8(lat)
    assert u8.shape == (2, 4, 128, 256, 3) and u8.dtype == np.uint8
    assert np.abs(u8[0].astype(np.int16) - ref.astype(np.int16)).mean() < 0.5


def test_vae_encode_moments_and_training_batch_from_pixels(vae_pair):
    """The training front from PIXELS (train_svd.py:846-931): `VAEHIP.encode_moments` (mean | clamped logvar of latent_dist)
    against the fp32 chain, and `train.batch_from_pixels` — sample() of the clip's frames x scaling_factor, rotated latent
    noise, the noise-augmented first frame's sample as conditioning, CLIP of the past observation, EDM input scaling —
    against the same composition on the oracle with the SAME random draws."""
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection

    from wiw_amd import train as T
    from wiw_amd.pipeline import action_ids_idx_encode, rotate_latent_noise
    from wiw_amd.vae import HIPFrontend

    vae, sd = vae_pair
    nb, lpb = len(VCFG["block_out_channels"]), VCFG["layers_per_block"]
    Tn, H, W = 4, 128, 256
    px = torch.tanh(rnd(1, Tn, 3, H, W, seed=21))
    mean, logvar = vae.encode_moments(px[0])
    rmean, rlogvar = VO.vae_encode_moments(sd, px[0], nb, lpb)
    (mx1, r1), (mx2, r2) = rel(mean, rmean), rel(logvar, rlogvar)
    print(f"[parity] VAE encode_moments {Tn}x{H}x{W}: mean rms_rel={r1:.3e} logvar rms_rel={r2:.3e}")
    assert mean.shape == (Tn, 4, H // 8, W // 8) and r1 <= 3e-2 and r2 <= 3e-2
    assert torch.equal(mean.cpu(), vae.encode_mode(px[0]).cpu())          # the mean half is the serving path's mode(), bit for bit

    torch.manual_seed(0)
    clip = CLIPVisionModelWithProjection(CLIPVisionConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=1,
                                                          num_attention_heads=2, image_size=224, patch_size=32,
                                                          projection_dim=1024)).eval()
    fe = HIPFrontend(vae, clip, dtype=torch.float32, clip="torch")
    acts = np.array([[1, 2, 1, 3]])
    h, w = H // 8, W // 8
    draws = dict(vae_eps=rnd(Tn, 4, h, w, seed=22), cond_eps=rnd(1, 4, h, w, seed=23), pixel_noise=rnd(1, 3, H, W, seed=24),
                 latent_noise=rnd(1, Tn, 4, h, w, seed=25), sigma=1.7, cond_sigma=0.043)
    past = torch.tanh(rnd(1, 1, 3, H, W, seed=26))
    st = T.batch_from_pixels(fe, px, past, acts, draws=draws)
    sf = vae.scaling_factor
    lat = ((rmean + torch.exp(0.5 * rlogvar) * draws["vae_eps"]) * sf)[None]
    noise = rotate_latent_noise(draws["latent_noise"], acts)
    cm, cl = VO.vae_encode_moments(sd, px[0, 0:1] + draws["pixel_noise"] * 0.043, nb, lpb)
    cond = cm + torch.exp(0.5 * cl) * draws["cond_eps"]
    noisy = lat + noise * 1.7
    inp = torch.cat([noisy / (1.7 ** 2 + 1) ** 0.5, cond.unsqueeze(1).repeat(1, Tn, 1, 1, 1)], dim=2)
    (_, rt), (_, ri) = rel(st.target, lat), rel(st.unet_input, inp)
    print(f"[parity] batch_from_pixels: target rms_rel={rt:.3e} unet_input rms_rel={ri:.3e}")
    assert st.unet_input.shape == (1, Tn, 8, h, w) and rt <= 3e-2 and ri <= 3e-2
    assert st.sigma == 1.7 and st.timestep == pytest.approx(0.25 * math.log(1.7)) and st.ehs.shape == (1, 1, 1024)
    assert torch.equal(st.action_ids, torch.from_numpy(action_ids_idx_encode(acts)))
    assert st.added_time_ids.tolist() == [[7.0, 127.0, pytest.approx(0.043)]]
    # turn at frame 1 (action 2): the noise of frame 1 is frame 0's, rolled by W/16 latent columns (pipeline:750-786)
    assert torch.equal((st.noisy - st.target)[0, 1], torch.roll((st.noisy - st.target)[0, 0], w // 16, dims=-1)) or \
        torch.allclose((st.noisy - st.target)[0, 1], torch.roll((st.noisy - st.target)[0, 0], w // 16, dims=-1), atol=1e-5)
    # drawn, not injected: a seeded generator reproduces itself and differs from another seed
    a = T.batch_from_pixels(fe, px, past, acts, generator=torch.Generator().manual_seed(5))
    b = T.batch_from_pixels(fe, px, past, acts, generator=torch.Generator().manual_seed(5))
    c = T.batch_from_pixels(fe, px, past, acts, generator=torch.Generator().manual_seed(6))
    assert torch.equal(a.unet_input, b.unet_input) and a.sigma == b.sigma and not torch.equal(a.unet_input, c.unet_input)
