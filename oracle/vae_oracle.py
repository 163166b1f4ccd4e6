"""CPU / PyTorch oracle for the temporal VAE and the worker front end  --  TEST INFRASTRUCTURE, NOT PRODUCT.

Functional fp32 re-statement (weights = the diffusers state dict of `vae/diffusion_pytorch_model*.safetensors`) of:
  * `AutoencoderKLTemporalDecoder.encode(...).latent_dist.mode()`  dp/models/autoencoders/autoencoder_kl_temporal_decoder.py:317-341,
    `Encoder.forward` dp/models/autoencoders/vae.py:139-184 (DownEncoderBlock2D / UNetMidBlock2D);
  * `TemporalDecoder.forward` autoencoder_kl_temporal_decoder.py:87-161 with `MidBlockTemporalDecoder` /
    `UpBlockTemporalDecoder` (dp/models/unets/unet_3d_blocks.py:930-1040): SpatioTemporalResBlock without time
    embedding, eps 1e-6 (spatial) / 1e-5 (temporal), AlphaBlender 'learned' with switch_spatial_to_temporal_mix;
  * `decode_latents` (pipeline:282-309): latents / scaling_factor, decode in chunks of `decode_chunk_size` frames.
dp/ = FTsvd/diffusers-private/diffusers/.

Pinned against outputs of the reference VAE run in the build container (oracle/make_golden.py ->
tests/golden/frontend_tiny.npz, checked by tests/test_frontend.py).  Only tests may import this module: the product
runs the VAE on the HIP kernels (world-in-world_amd/vae.py) and has no PyTorch / MIOpen route — that route was
measured at > 6 minutes per 576x1024x14 decode on a fresh MI355X box.
`TorchFrontend` (CLIP + this VAE through PyTorch, on any device) is the reference chain the end-to-end GPU test
compares the served uint8 frames with.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


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
    return vae_encode_moments(sd, x, n_blocks, layers_per_block)[0]


def vae_encode_moments(sd: SD, x: torch.Tensor, n_blocks: int = 4, layers_per_block: int = 2):
    """(B,3,H,W) in [-1,1] -> (mean, logvar) of `vae.encode(x).latent_dist` (DiagonalGaussianDistribution: logvar clamped to
    [-30, 20], dp/models/autoencoders/vae.py): what `.sample()` = mean + exp(0.5 logvar) eps uses (train_svd.py:86-95)."""
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
    L = m.shape[1] // 2
    return m[:, :L], torch.clamp(m[:, L:], -30.0, 20.0)


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
# the Frontend used by server.worker.SVDWorker
# ------------------------------------------------------------------------------------------------
def _product_frontend():
    import wiw_amd  # noqa: F401  (package alias)
    from wiw_amd import frontend
    return frontend


class TorchFrontend:
    """CLIP + VAE on a ROCm device through PyTorch (MIOpen / hipBLASLt underneath)."""

    def __init__(self, vae_state_dict, image_encoder, device="cuda:0", vae_dtype=torch.float32, dtype=torch.bfloat16,
                 scaling_factor: float = 0.18215, decode_chunk_size: Optional[int] = None,
                 block_out_channels: Sequence[int] = (128, 256, 512, 512), layers_per_block: int = 2):
        self.device = torch.device(device)
        spec = _product_frontend().vae_state_dict_spec(block_out_channels, layers_per_block)
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
    def encode(self, images: np.ndarray, image_noise: np.ndarray, noise_aug_strength: float, clip_images=None):
        x = torch.from_numpy(np.ascontiguousarray(images)).to(self.device, torch.float32)
        fe = _product_frontend()
        if clip_images is None:
            pix = fe.clip_preprocess(x)
        else:   # the CLIP branch runs on the un-resized images (pipeline:192-199)
            pix = torch.cat([fe.clip_preprocess(torch.from_numpy(np.ascontiguousarray(c))[None].to(self.device, torch.float32))
                             for c in clip_images])
        emb = self.image_encoder(pix.to(self.dtype)).image_embeds[:, None].float()   # (B,1,D)
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
