"""CLIP ViT image encoder on the HIP kernels.

Mirrors `StableVideoDiffusionPipeline._encode_image` (reference:
FTsvd/diffusers-private/diffusers/pipelines/stable_video_diffusion/pipeline_stable_video_diffusion.py:183-229) with the
third-party `transformers.CLIPVisionModelWithProjection` it calls (pinned 4.47 in FTsvd/train_svd.txt) restated as a
composition of this library's operators — same weights (a transformers state dict), same arguments:

    pixels in [-1,1] at the image's own size
      -> wiw_clip_preprocess   blur + bicubic(align_corners) to 224 + (x+1)/2 + mean/std + im2col         (clip.hip)
      -> patch GEMM (+ class / position embeddings as the residual operand)   -> pre-LayerNorm
      -> L x [ LN -> q|k GEMM, V^T GEMM (swapped operands) -> wiw_attn_small_bf16 (head_dim 80 for ViT-H/14)
               -> out-proj GEMM (+ residual) -> LN -> fc1 GEMM (+ GELU epilogue) -> fc2 GEMM (+ residual) ]
      -> post-LayerNorm -> visual projection (fp32 out)      -> image_embeds of the class-token rows

Layout: token-major bf16 [B * Sp, C] with Sp = 16 * ceil((1 + patches) / 16) rows per image (272 for 257 tokens): the
padding rows are zero inputs, are masked as keys and never read back.  V's bias is folded into the out-projection bias
(softmax rows sum to 1: P.(V + 1 b_v^T) = P.V + b_v  =>  b_o' = b_o + W_o b_v), so V^T comes from a bias-free GEMM with
swapped operands exactly as in the UNet's spatial attention.  Every kernel is deterministic and evaluates a row with
arithmetic that does not depend on the batch: an image's embedding is bit-identical whatever else is in the request.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from .frontend import CLIP_MEAN, CLIP_STD
from .hip import EPI_GELU, EPI_OUT_F32, EPI_QUICK_GELU, Hip, TiledW


def gaussian_taps(in_size: int, out_size: int):
    """Taps of `_resize_with_antialiasing`'s blur along one axis (pipeline:643-669): sigma = max((factor - 1) / 2, 0.001),
    kernel = max(int(4 sigma), 3) made odd, normalised."""
    fac = in_size / out_size
    sigma = max((fac - 1.0) / 2.0, 0.001)
    ks = int(max(2.0 * 2 * sigma, 3))
    ks = ks + 1 if ks % 2 == 0 else ks
    x = np.arange(ks, dtype=np.float32) - ks // 2
    g = np.exp(-(x ** 2) / np.float32(2 * sigma ** 2)).astype(np.float32)
    return (g / g.sum()).astype(np.float32).tolist()


class CLIPVisionHIP:
    def __init__(self, state_dict: Dict[str, "torch.Tensor | np.ndarray"], *, hidden_size: int, num_heads: int, num_layers: int,
                 intermediate_size: int, image_size: int = 224, patch_size: int = 14, projection_dim: int = 1024,
                 hidden_act: str = "gelu", layer_norm_eps: float = 1e-5, device="cuda:0", hip: Optional[Hip] = None):
        self.device = torch.device(device)
        self.hip = hip or Hip(self.device)
        self.C, self.heads, self.L, self.I = hidden_size, num_heads, num_layers, intermediate_size
        self.D = hidden_size // num_heads
        self.img, self.P, self.proj, self.eps = image_size, patch_size, projection_dim, layer_norm_eps
        if hidden_size % 64 or intermediate_size % 64 or self.D % 16 or self.D > 128 or image_size % patch_size:
            raise ValueError("CLIPVisionHIP: hidden / intermediate sizes must be multiples of 64, head_dim a multiple of 16 <= 128")
        if hidden_act not in ("gelu", "quick_gelu"):
            raise ValueError(f"unsupported CLIP activation {hidden_act!r}")
        self.act = EPI_GELU if hidden_act == "gelu" else EPI_QUICK_GELU
        self.grid = image_size // patch_size
        self.tokens = 1 + self.grid * self.grid
        self.Sp = -(-self.tokens // 16) * 16
        self.Kp = -(-(3 * patch_size * patch_size) // 64) * 64       # im2col K padded to the GEMM's K tile
        self.w: Dict[str, torch.Tensor] = {}
        self._prepare(state_dict)
        self._buf: Dict[tuple, tuple] = {}

    @classmethod
    def from_transformers(cls, model, device="cuda:0", hip: Optional[Hip] = None) -> "CLIPVisionHIP":
        """Build from a `transformers.CLIPVisionModelWithProjection` (weights are copied; the module is not kept)."""
        c = model.config
        return cls(model.state_dict(), hidden_size=c.hidden_size, num_heads=c.num_attention_heads,
                   num_layers=c.num_hidden_layers, intermediate_size=c.intermediate_size, image_size=c.image_size,
                   patch_size=c.patch_size, projection_dim=c.projection_dim, hidden_act=c.hidden_act,
                   layer_norm_eps=c.layer_norm_eps, device=device, hip=hip)

    # ------------------------------------------------------------------------------------------
    def _t(self, sd, name) -> torch.Tensor:
        v = sd[name]
        if isinstance(v, np.ndarray):
            v = torch.from_numpy(v)
        return v.detach().to(self.device, torch.float32)

    def _prepare(self, sd):
        w, bf, C = self.w, self.hip.dtype, self.C
        e = "vision_model.embeddings."
        pw = self._t(sd, e + "patch_embedding.weight").reshape(C, -1)          # (C, 3*P*P), flattened (c, ky, kx)
        w["patch.weight"] = torch.cat([pw, pw.new_zeros(C, self.Kp - pw.shape[1])], dim=1).to(bf).contiguous()
        pos = self._t(sd, e + "position_embedding.weight")                       # (tokens, C)
        emb = torch.zeros(self.Sp, C, device=self.device)
        emb[: self.tokens] = pos
        emb[0] += self._t(sd, e + "class_embedding")
        w["embed_rows"] = emb.to(bf).contiguous()                                # residual operand of the patch GEMM
        for n in ("pre_layrnorm", "post_layernorm"):
            w[n + ".weight"] = self._t(sd, f"vision_model.{n}.weight").contiguous()
            w[n + ".bias"] = self._t(sd, f"vision_model.{n}.bias").contiguous()
        for i in range(self.L):
            p = f"vision_model.encoder.layers.{i}."
            a = p + "self_attn."
            w[f"{i}.qk.weight"] = torch.cat([self._t(sd, a + "q_proj.weight"), self._t(sd, a + "k_proj.weight")]).to(bf).contiguous()
            w[f"{i}.qk.bias"] = torch.cat([self._t(sd, a + "q_proj.bias"), self._t(sd, a + "k_proj.bias")]).contiguous()
            w[f"{i}.v.weight"] = self._t(sd, a + "v_proj.weight").to(bf).contiguous()
            wo = self._t(sd, a + "out_proj.weight")
            w[f"{i}.o.weight"] = wo.to(bf).contiguous()
            w[f"{i}.o.bias"] = (self._t(sd, a + "out_proj.bias") + wo @ self._t(sd, a + "v_proj.bias")).contiguous()
            for n in ("layer_norm1", "layer_norm2"):
                w[f"{i}.{n}.weight"] = self._t(sd, p + n + ".weight").contiguous()
                w[f"{i}.{n}.bias"] = self._t(sd, p + n + ".bias").contiguous()
            for n in ("fc1", "fc2"):
                w[f"{i}.{n}.weight"] = self._t(sd, p + f"mlp.{n}.weight").to(bf).contiguous()
                w[f"{i}.{n}.bias"] = self._t(sd, p + f"mlp.{n}.bias").contiguous()
        w["proj.weight"] = self._t(sd, "visual_projection.weight").to(bf).contiguous()
        for k in list(w):   # static GEMM weights in the LDS-DMA tiled layout; `v.weight` is the A operand of a swapped GEMM
            if k.endswith(".weight") and w[k].dim() == 2 and w[k].dtype == bf and not k.endswith(".v.weight"):
                w[k] = TiledW(w[k])
        torch.cuda.synchronize(self.device)

    def _buffers(self, B: int, H0: int, W0: int):
        key = (B, H0, W0)
        if key not in self._buf:
            A = torch.zeros(B * self.Sp, self.Kp, dtype=self.hip.dtype, device=self.device)   # class / padding rows stay zero
            tmp = torch.empty(B * 3 * H0 * self.img, dtype=torch.float32, device=self.device)
            emb = self.w["embed_rows"].repeat(B, 1).contiguous()
            self._buf = {key: (A, tmp, emb)}                                                   # one geometry cached
        return self._buf[key]

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def embed_pixels(self, images: torch.Tensor) -> torch.Tensor:
        """images (B,3,H0,W0) fp32 in [-1,1] at their OWN size -> image_embeds (B, projection_dim) fp32."""
        hip, w, C, Sp = self.hip, self.w, self.C, self.Sp
        x = images.to(self.device, torch.float32).contiguous()
        B, _, H0, W0 = x.shape
        M = B * Sp
        A, tmp, emb = self._buffers(B, H0, W0)
        hip.clip_preprocess(x, B, H0, W0, self.img, self.P, gaussian_taps(W0, self.img), gaussian_taps(H0, self.img),
                            CLIP_MEAN, [1.0 / s for s in CLIP_STD], tmp, A, Sp, self.Kp)
        bf = self.hip.dtype
        h = torch.empty(M, C, dtype=bf, device=self.device)
        hip.gemm(A, w["patch.weight"], h, M=M, N=C, K=self.Kp, C1=self.Kp, res1=emb, ldr1=C, beta1=1.0)
        h = hip.layernorm(h, M, C, w["pre_layrnorm.weight"], w["pre_layrnorm.bias"], self.eps)
        a = torch.empty(M, C, dtype=bf, device=self.device)
        qk = torch.empty(M, 2 * C, dtype=bf, device=self.device)
        vt = torch.empty(C, M, dtype=bf, device=self.device)
        o = torch.empty(M, C, dtype=bf, device=self.device)
        g = torch.empty(M, self.I, dtype=bf, device=self.device)
        scale = float(self.D) ** -0.5
        for i in range(self.L):
            hip.layernorm(h, M, C, w[f"{i}.layer_norm1.weight"], w[f"{i}.layer_norm1.bias"], self.eps, out=a)
            hip.gemm(a, w[f"{i}.qk.weight"], qk, M=M, N=2 * C, K=C, C1=C, bias=w[f"{i}.qk.bias"])
            hip.gemm(w[f"{i}.v.weight"], a, vt, M=C, N=M, K=C, C1=C)            # V^T = Wv . a^T (bias folded into o.bias)
            hip.attn_small(qk, 2 * C, C, vt, M, o, C, B, self.tokens, Sp, self.heads, self.D, scale)
            h2 = torch.empty(M, C, dtype=bf, device=self.device)
            hip.gemm(o, w[f"{i}.o.weight"], h2, M=M, N=C, K=C, C1=C, bias=w[f"{i}.o.bias"], res1=h, ldr1=C, beta1=1.0)
            hip.layernorm(h2, M, C, w[f"{i}.layer_norm2.weight"], w[f"{i}.layer_norm2.bias"], self.eps, out=a)
            hip.gemm(a, w[f"{i}.fc1.weight"], g, M=M, N=self.I, K=C, C1=C, bias=w[f"{i}.fc1.bias"], epilogue=self.act)
            h = torch.empty(M, C, dtype=bf, device=self.device)
            hip.gemm(g, w[f"{i}.fc2.weight"], h, M=M, N=C, K=self.I, C1=self.I, bias=w[f"{i}.fc2.bias"], res1=h2, ldr1=C, beta1=1.0)
        # pooled output = post-LayerNorm of the class-token row, then the projection (all rows are processed: the class
        # rows are picked at the end — 271 of 272 rows are wasted work on a 1280 x 1024 GEMM, no gather kernel needed)
        hip.layernorm(h, M, C, w["post_layernorm.weight"], w["post_layernorm.bias"], self.eps, out=a)
        out = torch.empty(M, self.proj, dtype=torch.float32, device=self.device)
        hip.gemm(a, w["proj.weight"], out, M=M, N=self.proj, K=C, C1=C, epilogue=EPI_OUT_F32)
        return out.view(B, Sp, self.proj)[:, 0]

    def __call__(self, images: torch.Tensor) -> torch.Tensor:
        return self.embed_pixels(images)
