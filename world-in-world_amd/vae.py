"""Temporal VAE (AutoencoderKLTemporalDecoder) on the hand-written gfx950 kernels: the callers either side of
the denoising loop (SURVEY.md §8 rows a6 `_encode_vae_image`, a20 `decode_latents`; "next" rows f1 / f4).

Why it is not left to PyTorch-ROCm: on a fresh MI355X box the MIOpen convolutions behind `F.conv2d` / `F.conv3d`
need > 6 minutes for ONE 576x1024x14 decode (measured) against ~3 s for the 25-step denoise; the
same layers expressed on `wiw_gemm_bf16` (implicit-GEMM 3x3 / stride-2 / nearest-upsample / temporal convolutions),
the GroupNorm kernels and a row softmax run in a fraction of a second.

Reference modules restated (dp/ = FTsvd/diffusers-private/diffusers/):
  * `TemporalDecoder.forward`  dp/models/autoencoders/autoencoder_kl_temporal_decoder.py:87-161,
    `MidBlockTemporalDecoder` / `UpBlockTemporalDecoder`  dp/models/unets/unet_3d_blocks.py:930-1040,
    `SpatioTemporalResBlock` without time embedding  dp/models/resnet.py:686-716 (spatial eps 1e-6, temporal 1e-5),
    `AlphaBlender` 'learned' with switch_spatial_to_temporal_mix  dp/models/resnet.py:784-797;
  * `Encoder.forward`  dp/models/autoencoders/vae.py:139-184 (`DownEncoderBlock2D`, `UNetMidBlock2D`), `quant_conv`
    and `latent_dist.mode()`  autoencoder_kl_temporal_decoder.py:317-341;
  * the single-head (head_dim = C) self-attention of both mid blocks (legacy `AttnProcessor`).
Layout: token-major NHWC bf16 rows m = (frame*H + y)*W + x, fp32 accumulation and statistics; the reference runs
these modules in bf16 (`weight_dtype`), see pipeline:525-531, 615-620.  The parity checker is the fp32 functional
restatement in oracle/vae_oracle.py (itself pinned to the reference by tests/golden/frontend_tiny.npz).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from .frontend import vae_state_dict_spec
from .hip import A_CONV3X3, A_CONV3X3_S2P, A_CONV3X3_UP, A_CONV_T3, EPI_OUT_F32, Hip

CIN_PAD = 64   # conv_in input channels (4 latent / 3 pixel) are zero-padded to one 64-wide K tile


class VAEHIP:
    def __init__(self, state_dict: Dict[str, "torch.Tensor | np.ndarray"], device="cuda:0", hip: Optional[Hip] = None,
                 block_out_channels: Sequence[int] = (128, 256, 512, 512), layers_per_block: int = 2,
                 scaling_factor: float = 0.18215):
        self.device = torch.device(device)
        self.hip = hip or Hip(self.device)
        self.ch = list(block_out_channels)
        self.lpb = layers_per_block
        self.scaling_factor = scaling_factor
        spec = vae_state_dict_spec(block_out_channels, layers_per_block)
        missing = [k for k in spec if k not in state_dict]
        if missing:
            raise KeyError(f"VAE checkpoint is missing {len(missing)} tensors, e.g. {missing[:4]}")
        for k, shp in spec.items():
            if tuple(state_dict[k].shape) != tuple(shp):
                raise ValueError(f"{k}: expected shape {shp}, got {tuple(state_dict[k].shape)}")
        if any(c % 64 for c in self.ch):
            raise ValueError("VAEHIP needs block_out_channels that are multiples of 64 (one MFMA K tile)")
        self.w: Dict[str, torch.Tensor] = {}
        self.mix: Dict[str, float] = {}
        self._prepare(state_dict)

    # ------------------------------------------------------------------------------------------
    # weight re-layout (once, at load)
    # ------------------------------------------------------------------------------------------
    def _t(self, sd, name) -> torch.Tensor:
        v = sd[name]
        if isinstance(v, np.ndarray):
            v = torch.from_numpy(v)
        return v.to(self.device, torch.float32)

    def _prepare(self, sd):
        w, bf = self.w, self.hip.dtype

        def norm(p):
            w[p + ".weight"] = self._t(sd, p + ".weight").contiguous()
            w[p + ".bias"] = self._t(sd, p + ".bias").contiguous()

        def conv3(p, cin_pad=0):
            x = self._t(sd, p + ".weight").permute(0, 2, 3, 1)  # OIHW -> OHWI
            if cin_pad and x.shape[-1] < cin_pad:
                x = torch.cat([x, x.new_zeros(*x.shape[:-1], cin_pad - x.shape[-1])], dim=-1)
            w[p + ".weight"] = x.reshape(x.shape[0], -1).to(bf).contiguous()
            w[p + ".bias"] = self._t(sd, p + ".bias").contiguous()

        def convt(p):
            x = self._t(sd, p + ".weight")[:, :, :, 0, 0].permute(0, 2, 1)  # (O,I,3) -> (O,3,I)
            w[p + ".weight"] = x.reshape(x.shape[0], -1).to(bf).contiguous()
            w[p + ".bias"] = self._t(sd, p + ".bias").contiguous()

        def res2d(p):
            norm(p + ".norm1"); conv3(p + ".conv1"); norm(p + ".norm2"); conv3(p + ".conv2")
            if p + ".conv_shortcut.weight" in sd:
                w[p + ".conv_shortcut.weight"] = self._t(sd, p + ".conv_shortcut.weight")[:, :, 0, 0].to(bf).contiguous()
                w[p + ".conv_shortcut.bias"] = self._t(sd, p + ".conv_shortcut.bias").contiguous()

        def st_res(p):
            res2d(p + ".spatial_res_block")
            t = p + ".temporal_res_block"
            norm(t + ".norm1"); convt(t + ".conv1"); norm(t + ".norm2"); convt(t + ".conv2")
            # switch_spatial_to_temporal_mix: out = (1-s)*x_spatial + s*x_temporal with s = sigmoid(mix_factor)
            self.mix[p] = float(torch.sigmoid(self._t(sd, p + ".time_mixer.mix_factor")).item())

        def attn(p):
            norm(p + ".group_norm")
            for n in ("to_q", "to_k", "to_v", "to_out.0"):
                w[f"{p}.{n}.weight"] = self._t(sd, f"{p}.{n}.weight").to(bf).contiguous()
                w[f"{p}.{n}.bias"] = self._t(sd, f"{p}.{n}.bias").contiguous()

        ch, lpb = self.ch, self.lpb
        # ---- encoder
        conv3("encoder.conv_in", cin_pad=CIN_PAD)
        for i in range(len(ch)):
            for j in range(lpb):
                res2d(f"encoder.down_blocks.{i}.resnets.{j}")
            if i < len(ch) - 1:
                conv3(f"encoder.down_blocks.{i}.downsamplers.0.conv")
        res2d("encoder.mid_block.resnets.0"); attn("encoder.mid_block.attentions.0"); res2d("encoder.mid_block.resnets.1")
        norm("encoder.conv_norm_out")
        # conv_out (C -> 2L) followed by the 1x1 quant_conv (2L -> 2L) is one linear map; only the mean half
        # (latent_dist.mode()) is evaluated: W' = Wq[:L] . Wc,  b' = Wq[:L] . bc + bq[:L]   (composed in fp32)
        wc = self._t(sd, "encoder.conv_out.weight")                       # (2L, C, 3, 3)
        wq = self._t(sd, "quant_conv.weight")[:, :, 0, 0]                  # (2L, 2L)
        L = wq.shape[0] // 2
        self.latent_channels = L
        wcomp = torch.einsum("om,mchw->ochw", wq[:L], wc).permute(0, 2, 3, 1)
        w["encoder.mean_conv.weight"] = wcomp.reshape(L, -1).to(bf).contiguous()
        w["encoder.mean_conv.bias"] = (wq[:L] @ self._t(sd, "encoder.conv_out.bias") + self._t(sd, "quant_conv.bias")[:L]).contiguous()
        # both halves (mean | logvar) for `latent_dist.sample()`: the training front (train_svd.py:86-95)
        wfull = torch.einsum("om,mchw->ochw", wq, wc).permute(0, 2, 3, 1)
        w["encoder.moments_conv.weight"] = wfull.reshape(2 * L, -1).to(bf).contiguous()
        w["encoder.moments_conv.bias"] = (wq @ self._t(sd, "encoder.conv_out.bias") + self._t(sd, "quant_conv.bias")).contiguous()
        # ---- decoder
        conv3("decoder.conv_in", cin_pad=CIN_PAD)
        for j in range(lpb):
            st_res(f"decoder.mid_block.resnets.{j}")
        attn("decoder.mid_block.attentions.0")
        for i in range(len(ch)):
            for j in range(lpb + 1):
                st_res(f"decoder.up_blocks.{i}.resnets.{j}")
            if i < len(ch) - 1:
                conv3(f"decoder.up_blocks.{i}.upsamplers.0.conv")
        norm("decoder.conv_norm_out")
        conv3("decoder.conv_out")
        w["decoder.time_conv_out.weight"] = self._t(sd, "decoder.time_conv_out.weight")[:, :, :, 0, 0].contiguous()  # [co][ci][dt]
        w["decoder.time_conv_out.bias"] = self._t(sd, "decoder.time_conv_out.bias").contiguous()
        # static GEMM weights in the LDS-DMA tiled layout (hip.TiledW); `to_v` is the A operand of a swapped GEMM
        from .hip import TiledW
        for k in list(w):
            t_ = w[k]
            if (k.endswith(".weight") and torch.is_tensor(t_) and t_.dim() == 2 and t_.dtype == bf and t_.shape[1] % 64 == 0
                    and not k.endswith(".to_v.weight")):
                w[k] = TiledW(t_)
        torch.cuda.synchronize(self.device)

    # ------------------------------------------------------------------------------------------
    # blocks (token-major bf16 [M, C], M = frames * H * W)
    # ------------------------------------------------------------------------------------------
    def _empty(self, *shape, dtype=None):
        return torch.empty(shape, dtype=dtype or self.hip.dtype, device=self.device)

    def _conv3(self, p, x, M, Cin, H, W, mode=A_CONV3X3, **kw):
        Wt = self.w[p + ".weight"]
        N = Wt.shape[0]
        out = kw.pop("out", None)
        if out is None:
            out = self._empty(M, N)
        return self.hip.gemm(x, Wt, out, M=M, N=N, K=9 * Cin, C1=Cin, mode=mode, H=H, Wd=W, bias=self.w[p + ".bias"], **kw)

    def _res2d(self, p, x, Cin, M, H, W, eps=1e-6):
        """ResnetBlock2D without time embedding (resnet.py:320-373)."""
        hip, w = self.hip, self.w
        S = H * W
        Cout = w[p + ".conv1.weight"].shape[0]
        xn = hip.groupnorm(x, Cin, None, 0, M, S, w[p + ".norm1.weight"], w[p + ".norm1.bias"], eps, True)
        h = self._conv3(p + ".conv1", xn, M, Cin, H, W)
        del xn
        hn = hip.groupnorm(h, Cout, None, 0, M, S, w[p + ".norm2.weight"], w[p + ".norm2.bias"], eps, True)
        del h
        if p + ".conv_shortcut.weight" in w:
            sc = self._empty(M, Cout)
            hip.gemm(x, w[p + ".conv_shortcut.weight"], sc, M=M, N=Cout, K=Cin, C1=Cin, bias=w[p + ".conv_shortcut.bias"])
        else:
            sc = x
        return self._conv3(p + ".conv2", hn, M, Cout, H, W, res1=sc, ldr1=Cout, beta1=1.0), Cout

    def _st_res(self, p, x, Cin, M, H, W, T):
        """SpatioTemporalResBlock of the temporal decoder: temb None, merge 'learned', switched mix."""
        hip, w = self.hip, self.w
        S = H * W
        xs, C = self._res2d(p + ".spatial_res_block", x, Cin, M, H, W, 1e-6)
        t = p + ".temporal_res_block"
        xn = hip.groupnorm(xs, C, None, 0, M, T * S, w[t + ".norm1.weight"], w[t + ".norm1.bias"], 1e-5, True, clip=True)
        h = self._empty(M, C)
        hip.gemm(xn, w[t + ".conv1.weight"], h, M=M, N=C, K=3 * C, C1=C, mode=A_CONV_T3, H=H, Wd=W, T=T,
                 bias=w[t + ".conv1.bias"])
        hn = hip.groupnorm(h, C, None, 0, M, T * S, w[t + ".norm2.weight"], w[t + ".norm2.bias"], 1e-5, True, out=xn, clip=True)
        # (1-s)*xs + s*(xs + conv2(hn) + b) = xs + s*(acc + b)
        hip.gemm(hn, w[t + ".conv2.weight"], h, M=M, N=C, K=3 * C, C1=C, mode=A_CONV_T3, H=H, Wd=W, T=T,
                 bias=w[t + ".conv2.bias"], alpha=self.mix[p], res1=xs, ldr1=C, beta1=1.0)
        return h, C

    def _attn(self, p, x, C, frames, S):
        """GroupNorm -> single-head softmax(QK^T / sqrt(C)) V -> to_out + residual, per frame."""
        hip, w = self.hip, self.w
        if S % 64:
            raise ValueError(f"VAE attention needs H*W % 64 == 0 at the latent resolution, got {S}")
        M = frames * S
        t = hip.groupnorm(x, C, None, 0, M, S, w[p + ".group_norm.weight"], w[p + ".group_norm.bias"], 1e-6, False)
        q, k = self._empty(M, C), self._empty(M, C)
        hip.gemm(t, w[p + ".to_q.weight"], q, M=M, N=C, K=C, C1=C, bias=w[p + ".to_q.bias"])
        hip.gemm(t, w[p + ".to_k.weight"], k, M=M, N=C, K=C, C1=C, bias=w[p + ".to_k.bias"])
        o = self._empty(M, C)
        vt = self._empty(C, S)
        sc = self._empty(S, S, dtype=torch.float32)
        pm = self._empty(S, S)
        scale = 1.0 / math.sqrt(C)
        for f in range(frames):
            r = slice(f * S, (f + 1) * S)
            # V^T [C][S] by swapping the operands; its bias is added after P.V (softmax rows sum to 1)
            hip.gemm(w[p + ".to_v.weight"], t[r], vt, M=C, N=S, K=C, C1=C)
            hip.gemm(q[r], k[r], sc, M=S, N=S, K=C, C1=C, alpha=scale, epilogue=EPI_OUT_F32)
            hip.softmax_rows(sc, S, S, S, pm, S)
            hip.gemm(pm, vt, o[r], M=S, N=C, K=S, C1=S, bias=w[p + ".to_v.bias"])
        out = self._empty(M, C)
        return hip.gemm(o, w[p + ".to_out.0.weight"], out, M=M, N=C, K=C, C1=C, bias=w[p + ".to_out.0.bias"],
                        res1=x, ldr1=C, beta1=1.0)

    # ------------------------------------------------------------------------------------------
    # decode / encode
    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def decode_frames(self, z: torch.Tensor) -> torch.Tensor:
        """z: fp32 (T, L, h, w) latents of ONE clip, already divided by scaling_factor (the reference decodes
        `decode_chunk_size = num_frames` frames per call, pipeline:291-302).  Returns fp32 (T, 3, 8h, 8w)."""
        hip, w, ch, lpb = self.hip, self.w, self.ch, self.lpb
        z = z.to(self.device, torch.float32).contiguous()
        T, L, H, W = z.shape
        M = T * H * W
        x = hip.nchw_to_nhwc(z, T, L, H * W, 1.0, CIN_PAD, self._empty(M, CIN_PAD))
        h = self._conv3("decoder.conv_in", x, M, CIN_PAD, H, W)
        C = ch[-1]
        h, C = self._st_res("decoder.mid_block.resnets.0", h, C, M, H, W, T)
        for j in range(1, lpb):
            if j == 1:
                h = self._attn("decoder.mid_block.attentions.0", h, C, T, H * W)
            h, C = self._st_res(f"decoder.mid_block.resnets.{j}", h, C, M, H, W, T)
        for i in range(len(ch)):
            for j in range(lpb + 1):
                h, C = self._st_res(f"decoder.up_blocks.{i}.resnets.{j}", h, C, M, H, W, T)
            p = f"decoder.up_blocks.{i}.upsamplers.0.conv"
            if p + ".weight" in w:
                H, W = 2 * H, 2 * W
                M = T * H * W
                h = self._conv3(p, h, M, C, H, W, mode=A_CONV3X3_UP)
        hn = hip.groupnorm(h, C, None, 0, M, H * W, w["decoder.conv_norm_out.weight"], w["decoder.conv_norm_out.bias"],
                           1e-6, True)
        del h
        y = self._empty(M, 4, dtype=torch.float32)
        self._conv3("decoder.conv_out", hn, M, C, H, W, out=y, ldo=4, epilogue=EPI_OUT_F32)
        frames = self._empty(T, 3, H, W, dtype=torch.float32)
        hip.vae_time_conv_out(y, 4, w["decoder.time_conv_out.weight"], w["decoder.time_conv_out.bias"], T, T, H * W, frames)
        return frames

    @torch.no_grad()
    def encode_mode(self, x: torch.Tensor) -> torch.Tensor:
        """x: fp32 (B, 3, H, W) in [-1, 1] -> latent_dist.mode() fp32 (B, L, H/8, W/8), NOT multiplied by
        scaling_factor (pipeline:239)."""
        return self._encode(x, "encoder.mean_conv", self.latent_channels)

    @torch.no_grad()
    def encode_moments(self, x: torch.Tensor):
        """x: fp32 (B, 3, H, W) in [-1, 1] -> (mean, logvar) of `vae.encode(x).latent_dist`, fp32 (B, L, H/8, W/8) each, logvar
        clamped to [-30, 20] as DiagonalGaussianDistribution does (dp/models/autoencoders/vae.py): what `.sample()` needs
        (train_svd.py:91)."""
        L = self.latent_channels
        m = self._encode(x, "encoder.moments_conv", 2 * L)
        return m[:, :L].contiguous(), torch.clamp(m[:, L:], -30.0, 20.0).contiguous()

    def _encode(self, x: torch.Tensor, head: str, L: int) -> torch.Tensor:
        hip, w, ch, lpb = self.hip, self.w, self.ch, self.lpb
        x = x.to(self.device, torch.float32).contiguous()
        B, Cx, H, W = x.shape
        if H % (8 * 8) or W % (8 * 8):
            raise ValueError("VAEHIP.encode_* needs H and W that are multiples of 64")
        M = B * H * W
        h = hip.nchw_to_nhwc(x, B, Cx, H * W, 1.0, CIN_PAD, self._empty(M, CIN_PAD))
        h = self._conv3("encoder.conv_in", h, M, CIN_PAD, H, W)
        C = ch[0]
        for i in range(len(ch)):
            for j in range(lpb):
                h, C = self._res2d(f"encoder.down_blocks.{i}.resnets.{j}", h, C, M, H, W)
            p = f"encoder.down_blocks.{i}.downsamplers.0.conv"
            if p + ".weight" in w:
                H, W = H // 2, W // 2
                M = B * H * W
                h = self._conv3(p, h, M, C, H, W, mode=A_CONV3X3_S2P)
        h, C = self._res2d("encoder.mid_block.resnets.0", h, C, M, H, W)
        h = self._attn("encoder.mid_block.attentions.0", h, C, B, H * W)
        h, C = self._res2d("encoder.mid_block.resnets.1", h, C, M, H, W)
        hn = hip.groupnorm(h, C, None, 0, M, H * W, w["encoder.conv_norm_out.weight"], w["encoder.conv_norm_out.bias"],
                           1e-6, True)
        y = self._empty(M, L, dtype=torch.float32)
        self._conv3(head, hn, M, C, H, W, out=y, ldo=L, epilogue=EPI_OUT_F32)
        return y.reshape(B, H, W, L).permute(0, 3, 1, 2).contiguous()

    @torch.no_grad()
    def decode(self, latents: torch.Tensor) -> torch.Tensor:
        """decode_latents (pipeline:282-309): latents fp32 (B, T, L, h, w) -> frames fp32 (B, T, 3, 8h, 8w)."""
        z = latents.to(self.device, torch.float32) / self.scaling_factor
        return torch.stack([self.decode_frames(z[b]) for b in range(z.shape[0])])


class HIPFrontend:
    """`server.worker.Frontend` on the HIP kernels: temporal VAE (`VAEHIP`) and the CLIP image encoder
    (`clip.CLIPVisionHIP`, built from the weights of the `transformers` module the reference uses, pipeline:183-229)."""

    def __init__(self, vae: VAEHIP, image_encoder, dtype=None, device_io: bool = False, clip: str = "hip"):
        """image_encoder: a `transformers.CLIPVisionModelWithProjection` (its weights are taken).
        clip = "hip" (product): the encoder runs on the HIP kernels and raises if its geometry is unsupported;
        clip = "torch": the module itself runs on PyTorch-ROCm (checker / reduced-width test models only).
        device_io: `encode` returns DEVICE tensors (and `decode*` accept them): the latents stay in HBM between
        encode -> denoise -> decode instead of bouncing through host numpy."""
        from .frontend import clip_preprocess
        self.device_io = device_io
        self._clip_preprocess = clip_preprocess
        self.vae = vae
        self.device = vae.device
        self.dtype = dtype or vae.hip.dtype
        self.clip_hip = None
        self.image_encoder = None
        if clip == "hip":
            from .clip import CLIPVisionHIP
            self.clip_hip = CLIPVisionHIP.from_transformers(image_encoder, self.device, hip=vae.hip)
        elif clip == "torch":
            self.image_encoder = image_encoder.to(self.device, self.dtype).eval() if image_encoder is not None else None
        else:
            raise ValueError("clip must be 'hip' or 'torch'")

    def _embed(self, x: torch.Tensor, clip_images) -> torch.Tensor:
        """(B,1,D) fp32 CLIP image embeddings; `clip_images` = per-candidate un-resized images (pipeline:192-199) or None."""
        groups = [x] if clip_images is None else [torch.from_numpy(np.ascontiguousarray(c))[None].to(self.device, torch.float32)
                                                   for c in clip_images]
        if clip_images is not None and len({tuple(g.shape) for g in groups}) == 1:
            groups = [torch.cat(groups)]                     # same size (the normal case): one batched pass
        if self.clip_hip is not None:
            emb = torch.cat([self.clip_hip(g) for g in groups])
        else:
            emb = torch.cat([self.image_encoder(self._clip_preprocess(g).to(self.dtype)).image_embeds for g in groups])
        return emb[:, None].float()

    @torch.no_grad()
    def encode(self, images: np.ndarray, image_noise: np.ndarray, noise_aug_strength: float, clip_images=None):
        x = torch.from_numpy(np.ascontiguousarray(images)).to(self.device, torch.float32)
        emb = self._embed(x, clip_images)
        xn = x + noise_aug_strength * torch.from_numpy(np.ascontiguousarray(image_noise)).to(self.device, torch.float32)
        lat = self.vae.encode_mode(xn)
        if self.device_io:
            return lat, emb
        return lat.cpu().numpy(), emb.cpu().numpy()

    @torch.no_grad()
    def decode(self, latents) -> np.ndarray:
        z = latents if torch.is_tensor(latents) else torch.from_numpy(np.ascontiguousarray(latents))
        return self.vae.decode(z).cpu().numpy()

    @torch.no_grad()
    def decode_uint8(self, latents: np.ndarray) -> np.ndarray:
        """decode + the frame quantisation of `frames_to_pil` on the device: (x/2 + 0.5).clamp(0, 1) * 255, round half
        to even, uint8, HWC (image_processor.py:133-147) — the same fp32 operations in the same order, so the bytes
        are identical; 25 MB instead of 99 MB cross PCIe per candidate and the host skips a 0.3-0.7 s numpy pass."""
        z = latents if torch.is_tensor(latents) else torch.from_numpy(np.ascontiguousarray(latents))
        out = []
        for b in range(z.shape[0]):          # per clip: bounds the fp32 frame buffer to one candidate
            out.append(frames_to_uint8_device(self.vae.decode(z[b:b + 1])[0]).cpu().numpy())
        return np.stack(out)


def frames_to_uint8_device(frames: torch.Tensor) -> torch.Tensor:
    """(T,3,H,W) fp32 in [-1,1] on the device -> (T,H,W,3) uint8: `server.plumbing.frames_to_pil` without the host."""
    u8 = torch.clamp(frames / 2.0 + 0.5, 0.0, 1.0).mul_(255.0).round_().to(torch.uint8)
    return u8.permute(0, 2, 3, 1).contiguous()
