"""Host-side pieces around the VAE / CLIP that the product needs whatever runs the VAE: the diffusers state-dict schema
of `AutoencoderKLTemporalDecoder` (validation, seeded random init for tests and bring-up) and the CLIP image
preprocessing of `_encode_image` (pipeline:183-229): `_resize_with_antialiasing` (pipeline:643-746) to 224x224 and the
CLIP normalisation.  The image encoder itself is the third-party `transformers.CLIPVisionModelWithProjection`, as in
the reference.  The VAE runs on the HIP kernels (vae.py); its fp32 PyTorch restatement — the parity checker, pinned
to the reference by tests/golden/frontend_tiny.npz — lives in oracle/vae_oracle.py (test infrastructure).
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
