"""PyTorch side of the worker: CLIP image embedding + VAE encode (conditioning) and the temporal VAE
decoder (frames).  north_star keeps these in PyTorch-ROCm ("only for tensor plumbing and the VAE
encode/decode"); they are rows a5, a6, a20 of SURVEY.md §8(a) and the "next" rows (f1, f4).

Functional re-statement (weights = the diffusers state dict of `vae/diffusion_pytorch_model*.safetensors`)
of the reference modules:
  * `AutoencoderKLTemporalDecoder.encode(...).latent_dist.mode()`  dp/models/autoencoders/autoencoder_kl_temporal_decoder.py:317-341,
    `Encoder.forward` dp/models/autoencoders/vae.py:139-184 (DownEncoderBlock2D / UNetMidBlock2D);
  * `TemporalDecoder.forward` autoencoder_kl_temporal_decoder.py:87-161 with `MidBlockTemporalDecoder` /
    `UpBlockTemporalDecoder` (dp/models/unets/unet_3d_blocks.py:930-1040): SpatioTemporalResBlock without time
    embedding, eps 1e-6 (spatial) / 1e-5 (temporal), AlphaBlender 'learned' with switch_spatial_to_temporal_mix;
  * `_encode_image` (pipeline:183-229): `_resize_with_antialiasing` (pipeline:643-746) to 224x224, CLIP
    normalisation, `transformers.CLIPVisionModelWithProjection` (third-party, as in the reference);
  * `decode_latents` (pipeline:282-309): latents / scaling_factor, decode in chunks of `decode_chunk_size` frames.
dp/ = FTsvd/diffusers-private/diffusers/.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


# ------------------------------------------------------------------------------------------------
# schema (for validation and seeded random init in tests)
# ------------------------------------------------------------------------------------------------
def vae_state_dict_spec(block_out_channels: Sequence[int] = (128, 256, 512, 512), layers_per_block: int = 2,
                        latent_channels: int = 4) -> "OrderedDict[str, Tuple[int, ...]]":
    spec: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    ch = list(block_out_channels)

    def conv(p, cin, cout, k=3):
        spec[p + ".weight"] = (cout, cin, k, k)
        spec[p + ".bias"] = (cout,)

    def norm(p, c):
        spec[p + ".weight"] = (c,)
        spec[p + ".bias"] = (c,)

    def res2d(p, cin, cout):
        norm(p + ".norm1", cin); conv(p + ".conv1", cin, cout); norm(p + ".norm2", cout); conv(p + ".conv2", cout, cout)
        if cin != cout:
            conv(p + ".conv_shortcut", cin, cout, 1)

    def attn(p, c):
        norm(p + ".group_norm", c)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            spec[f"{p}.{n}.weight"] = (c, c)
            spec[f"{p}.{n}.bias"] = (c,)

    def st_res(p, cin, cout):
        res2d(p + ".spatial_res_block", cin, cout)
        t = p + ".temporal_res_block"
        norm(t + ".norm1", cout)
        spec[t + ".conv1.weight"] = (cout, cout, 3, 1, 1); spec[t + ".conv1.bias"] = (cout,)
        norm(t + ".norm2", cout)
        spec[t + ".conv2.weight"] = (cout, cout, 3, 1, 1); spec[t + ".conv2.bias"] = (cout,)
        spec[p + ".time_mixer.mix_factor"] = (1,)

    conv("encoder.conv_in", 3, ch[0])
    cout = ch[0]
    for i, c in enumerate(ch):
        cin, cout = cout, c
        for j in range(layers_per_block):
            res2d(f"encoder.down_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout)
        if i < len(ch) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", cout, cout)
    res2d("encoder.mid_block.resnets.0", ch[-1], ch[-1])
    attn("encoder.mid_block.attentions.0", ch[-1])
    res2d("encoder.mid_block.resnets.1", ch[-1], ch[-1])
    norm("encoder.conv_norm_out", ch[-1])
    conv("encoder.conv_out", ch[-1], 2 * latent_channels)
    conv("quant_conv", 2 * latent_channels, 2 * latent_channels, 1)
    conv("decoder.conv_in", latent_channels, ch[-1])
    for j in range(layers_per_block):
        st_res(f"decoder.mid_block.resnets.{j}", ch[-1], ch[-1])
    attn("decoder.mid_block.attentions.0", ch[-1])
    rch = list(reversed(ch))
    cout = rch[0]
    for i, c in enumerate(rch):
        cin, cout = cout, c
        for j in range(layers_per_block + 1):
            st_res(f"decoder.up_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout)
        if i < len(rch) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", cout, cout)
    norm("decoder.conv_norm_out", ch[0])
    conv("decoder.conv_out", ch[0], 3)
    spec["decoder.time_conv_out.weight"] = (3, 3, 3, 1, 1)
    spec["decoder.time_conv_out.bias"] = (3,)
    return spec


def vae_random_state_dict(seed: int = 0, **cfg) -> Dict[str, np.ndarray]:
    rs = np.random.RandomState(seed)
    spec = vae_state_dict_spec(**cfg)
    sd = {}
    for name, shape in spec.items():
        if name.endswith("mix_factor"):
            v = rs.uniform(-1.0, 1.0, size=shape)
        elif len(shape) == 1 and "norm" in name:
            v = (1.0 + 0.1 * rs.standard_normal(shape)) if name.endswith(".weight") else 0.1 * rs.standard_normal(shape)
        else:
            wshape = spec.get(name[:-5] + ".weight", shape) if name.endswith(".bias") else shape
            fan_in = int(np.prod(wshape[1:])) if len(wshape) > 1 else 1
            v = rs.uniform(-1.0, 1.0, size=shape) / np.sqrt(max(fan_in, 1))
        sd[name] = v.astype(np.float32)
    return sd


# ------------------------------------------------------------------------------------------------
# blocks
# ------------------------------------------------------------------------------------------------
def _gn(sd: SD, p: str, x, eps):
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps)


def _res2d(sd: SD, p: str, x, eps=1e-6):
    """ResnetBlock2D without time embedding (resnet.py:320-373)."""
    h = F.conv2d(F.silu(_gn(sd, p + ".norm1", x, eps)), sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    h = F.conv2d(F.silu(_gn(sd, p + ".norm2", h, eps)), sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    if p + ".conv_shortcut.weight" in sd:
        x = F.conv2d(x, sd[p + ".conv_shortcut.weight"], sd[p + ".conv_shortcut.bias"])
    return x + h


def _attn2d(sd: SD, p: str, x):
    """Single-head self-attention over the H*W sites with GroupNorm and residual (legacy `AttnProcessor`,
    attention_processor.py; heads = C / attention_head_dim = 1 in both VAE mid blocks)."""
    n, c, h, w = x.shape
    t = _gn(sd, p + ".group_norm", x, 1e-6).reshape(n, c, h * w).transpose(1, 2)
    q = F.linear(t, sd[p + ".to_q.weight"], sd[p + ".to_q.bias"])
    k = F.linear(t, sd[p + ".to_k.weight"], sd[p + ".to_k.bias"])
    v = F.linear(t, sd[p + ".to_v.weight"], sd[p + ".to_v.bias"])
    o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
    o = F.linear(o, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])
    return o.transpose(1, 2).reshape(n, c, h, w) + x


def _st_res(sd: SD, p: str, x, T: int):
    """SpatioTemporalResBlock as configured in the temporal decoder (temb None, merge 'learned', switched)."""
    xs = _res2d(sd, p + ".spatial_res_block", x, 1e-6)
    n, c, h, w = xs.shape
    x5 = xs.reshape(n // T, T, c, h, w).permute(0, 2, 1, 3, 4)
    t = p + ".temporal_res_block"
    g = F.conv3d(F.silu(_gn(sd, t + ".norm1", x5, 1e-5)), sd[t + ".conv1.weight"], sd[t + ".conv1.bias"], padding=(1, 0, 0))
    g = F.conv3d(F.silu(_gn(sd, t + ".norm2", g, 1e-5)), sd[t + ".conv2.weight"], sd[t + ".conv2.bias"], padding=(1, 0, 0))
    xt = x5 + g
    a = 1.0 - torch.sigmoid(sd[p + ".time_mixer.mix_factor"]).reshape(()).to(x.dtype)  # switch_spatial_to_temporal_mix
    out = a * x5 + (1.0 - a) * xt
    return out.permute(0, 2, 1, 3, 4).reshape(n, c, h, w)


def vae_encode_mode(sd: SD, x: torch.Tensor, n_blocks: int = 4, layers_per_block: int = 2) -> torch.Tensor:
    """(B,3,H,W) in [-1,1] -> latent_dist.mode() (B,4,H/8,W/8); NOT multiplied by scaling_factor (pipeline:239)."""
    h = F.conv2d(x, sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=1)
    for i in range(n_blocks):
        for j in range(layers_per_block):
            h = _res2d(sd, f"encoder.down_blocks.{i}.resnets.{j}", h)
        p = f"encoder.down_blocks.{i}.downsamplers.0.conv"
        if p + ".weight" in sd:  # Downsample2D(padding=0): pad (0,1,0,1) then stride-2 conv (downsampling.py:132-150)
            h = F.conv2d(F.pad(h, (0, 1, 0, 1)), sd[p + ".weight"], sd[p + ".bias"], stride=2)
    h = _res2d(sd, "encoder.mid_block.resnets.0", h)
    h = _attn2d(sd, "encoder.mid_block.attentions.0", h)
    h = _res2d(sd, "encoder.mid_block.resnets.1", h)
    h = F.silu(_gn(sd, "encoder.conv_norm_out", h, 1e-6))
    h = F.conv2d(h, sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"], padding=1)
    m = F.conv2d(h, sd["quant_conv.weight"], sd["quant_conv.bias"])
    return m[:, : m.shape[1] // 2]


def vae_decode(sd: SD, z: torch.Tensor, num_frames: int, n_blocks: int = 4, layers_per_block: int = 2) -> torch.Tensor:
    """(B*T,4,h,w) latents (already divided by scaling_factor) -> (B*T,3,8h,8w)."""
    T = num_frames
    h = F.conv2d(z, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    h = _st_res(sd, "decoder.mid_block.resnets.0", h, T)
    for j in range(1, layers_per_block):
        h = _attn2d(sd, "decoder.mid_block.attentions.0", h) if j == 1 else h
        h = _st_res(sd, f"decoder.mid_block.resnets.{j}", h, T)
    for i in range(n_blocks):
        for j in range(layers_per_block + 1):
            h = _st_res(sd, f"decoder.up_blocks.{i}.resnets.{j}", h, T)
        p = f"decoder.up_blocks.{i}.upsamplers.0.conv"
        if p + ".weight" in sd:
            h = F.conv2d(F.interpolate(h, scale_factor=2.0, mode="nearest"), sd[p + ".weight"], sd[p + ".bias"], padding=1)
    h = F.silu(_gn(sd, "decoder.conv_norm_out", h, 1e-6))
    h = F.conv2d(h, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)
    n, c, hh, ww = h.shape
    h5 = h.reshape(n // T, T, c, hh, ww).permute(0, 2, 1, 3, 4)
    h5 = F.conv3d(h5, sd["decoder.time_conv_out.weight"], sd["decoder.time_conv_out.bias"], padding=(1, 0, 0))
    return h5.permute(0, 2, 1, 3, 4).reshape(n, c, hh, ww)


# ------------------------------------------------------------------------------------------------
# CLIP preprocessing
# ------------------------------------------------------------------------------------------------
def _gaussian(window: int, sigma: float, dtype) -> torch.Tensor:
    x = torch.arange(window, dtype=dtype) - window // 2
    if window % 2 == 0:
        x = x + 0.5
    g = torch.exp(-x.pow(2.0) / (2 * sigma ** 2))
    return g / g.sum()


def resize_with_antialiasing(x: torch.Tensor, size=(224, 224)) -> torch.Tensor:
    """`_resize_with_antialiasing` (pipeline:643-669): separable Gaussian blur (sigma = (factor-1)/2, 2-sigma
    odd kernel, reflect padding) then bicubic interpolation with align_corners=True."""
    h, w = x.shape[-2:]
    fac = (h / size[0], w / size[1])
    sig = (max((fac[0] - 1.0) / 2.0, 0.001), max((fac[1] - 1.0) / 2.0, 0.001))
    ks = [int(max(2.0 * 2 * s, 3)) for s in sig]
    ks = [k + 1 if k % 2 == 0 else k for k in ks]
    kx = _gaussian(ks[1], sig[1], x.dtype).tolist()
    ky = _gaussian(ks[0], sig[0], x.dtype).tolist()
    px, py = ks[1] - 1, ks[0] - 1
    # the separable depthwise blur as explicit tap sums (3-7 taps): on ROCm the grouped fp32 F.conv2d of the reference
    # falls to MIOpen's naive kernel (2.3 ms per call at 576x1024, 18 ms per request); same taps, same order
    xp = F.pad(x, (px // 2, px - px // 2, 0, 0), mode="reflect")
    out = kx[0] * xp[..., 0:w]
    for k in range(1, ks[1]):
        out = out + kx[k] * xp[..., k:k + w]
    yp = F.pad(out, (0, 0, py // 2, py - py // 2), mode="reflect")
    out = ky[0] * yp[..., 0:h, :]
    for k in range(1, ks[0]):
        out = out + ky[k] * yp[..., k:k + h, :]
    return F.interpolate(out, size=size, mode="bicubic", align_corners=True)


def clip_preprocess(images: torch.Tensor) -> torch.Tensor:
    """images (B,3,H,W) in [-1,1] -> CLIP pixel_values (B,3,224,224)  (pipeline:197-212)."""
    x = (resize_with_antialiasing(images, (224, 224)) + 1.0) / 2.0
    mean = torch.tensor(CLIP_MEAN, dtype=x.dtype, device=x.device).reshape(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD, dtype=x.dtype, device=x.device).reshape(1, 3, 1, 1)
    return (x - mean) / std


# ------------------------------------------------------------------------------------------------
# the Frontend used by server.worker.SVDWorker
# ------------------------------------------------------------------------------------------------
class TorchFrontend:
    """CLIP + VAE on a ROCm device through PyTorch (MIOpen / hipBLASLt underneath)."""

    def __init__(self, vae_state_dict, image_encoder, device="cuda:0", vae_dtype=torch.float32, dtype=torch.bfloat16,
                 scaling_factor: float = 0.18215, decode_chunk_size: Optional[int] = None,
                 block_out_channels: Sequence[int] = (128, 256, 512, 512), layers_per_block: int = 2):
        self.device = torch.device(device)
        spec = vae_state_dict_spec(block_out_channels, layers_per_block)
        missing = [k for k in spec if k not in vae_state_dict]
        if missing:
            raise KeyError(f"VAE checkpoint is missing {len(missing)} tensors, e.g. {missing[:4]}")
        self.vae_dtype, self.dtype = vae_dtype, dtype
        to_t = lambda v: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v)  # noqa: E731
        self.sd_enc = {k: to_t(v).to(self.device, vae_dtype) for k, v in vae_state_dict.items()
                       if k.startswith(("encoder.", "quant_conv."))}   # encode in fp32 (force_upcast, pipeline:525-527)
        self.sd_dec = {k: to_t(v).to(self.device, dtype) for k, v in vae_state_dict.items() if k.startswith("decoder.")}
        self.image_encoder = image_encoder.to(self.device, dtype).eval() if image_encoder is not None else None
        self.scaling_factor = scaling_factor
        self.decode_chunk_size = decode_chunk_size
        self.nb, self.lpb = len(block_out_channels), layers_per_block

    @torch.no_grad()
    def encode(self, images: np.ndarray, image_noise: np.ndarray, noise_aug_strength: float):
        x = torch.from_numpy(np.ascontiguousarray(images)).to(self.device, torch.float32)
        emb = self.image_encoder(clip_preprocess(x).to(self.dtype)).image_embeds[:, None].float()   # (B,1,D)
        xn = x + noise_aug_strength * torch.from_numpy(np.ascontiguousarray(image_noise)).to(self.device, torch.float32)
        lat = vae_encode_mode(self.sd_enc, xn.to(self.vae_dtype), self.nb, self.lpb).float()
        return lat.cpu().numpy(), emb.cpu().numpy()

    @torch.no_grad()
    def decode(self, latents: np.ndarray) -> np.ndarray:
        z = torch.from_numpy(np.ascontiguousarray(latents)).to(self.device, self.dtype)
        B, T = z.shape[:2]
        z = z.flatten(0, 1) / self.scaling_factor
        chunk = self.decode_chunk_size or T
        outs = [vae_decode(self.sd_dec, z[i:i + chunk], z[i:i + chunk].shape[0], self.nb, self.lpb)
                for i in range(0, z.shape[0], chunk)]
        fr = torch.cat(outs).float()
        return fr.reshape(B, T, *fr.shape[1:]).cpu().numpy()
