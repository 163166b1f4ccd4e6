"""Spatio-temporal UNet forward on the HIP kernels (host orchestration).

Mirrors `UNetSpatioTemporalConditionModel.forward` of the reference
(FTsvd/diffusers-private/diffusers/models/unets/unet_spatio_temporal_condition.py:402-575,
`micro_cond` action strategy) — same weights (diffusers state dict), same arguments — but laid out
for MI355X:

  * activations are token-major bf16 [B*T*H*W, C] (NHWC); the (B*T,S,C) <-> (B*S,T,C) and
    (BT,C,H,W) <-> (B,C,T,H,W) permutes of the reference (attention.py:720-722, resnet.py:698-715)
    are address arithmetic inside the kernels, never materialised;
  * every conv / linear is one call of the MFMA GEMM kernel with fused bias / time-embedding /
    residual / GEGLU / AlphaBlender epilogues; GroupNorm+SiLU and LayerNorm(+pre-add) are the only
    standalone normalisation passes;
  * the two single-key cross-attentions per transformer layer are evaluated in closed form
    (softmax over one key == 1  =>  attn2(x) = to_out(to_v(ctx))): one [C] vector
    per CFG-batch item, computed once per request and added inside the following LayerNorm kernel;
  * batch >= 2 follows the contract "candidate i == reference B=1 run on candidate i" (§9.2).
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np
import torch

from .config import UNetConfig
from .hip import (A_CONV3X3, A_CONV3X3_S2, A_CONV3X3_UP, A_CONV_T3, A_DENSE, EPI_GEGLU, EPI_OUT_F32, EPI_SILU,
                  FFN_C, FFN_CHUNK, FFN_HIDDEN, GEGLU_TILE, K_CMAJOR, K_HALO32, Hip, TiledW, tile_weight)
from .weights import validate_state_dict

CIN_PAD = 64  # conv_in input channels padded 8 -> 64 so it runs on the MFMA conv kernel
ATTN_PRESCALE = math.log2(math.e) / math.sqrt(64.0)   # folded into the spatial to_q rows: wiw_attn_spatial_ps_bf16 takes exp2(Q.K)


def sinusoid(t: np.ndarray, dim: int) -> np.ndarray:
    """get_timestep_embedding (dp/models/embeddings.py:27-78), flip_sin_to_cos=True, shift 0: [cos | sin]."""
    half = dim // 2
    freq = np.exp(-math.log(10000.0) * np.arange(half, dtype=np.float32) / np.float32(half)).astype(np.float32)
    arg = t.astype(np.float32)[:, None] * freq[None, :]
    return np.concatenate([np.cos(arg), np.sin(arg)], axis=-1).astype(np.float32)


def action_features(action_ids: np.ndarray) -> np.ndarray:
    """ActionEmbedder_ features (dp/models/embeddings.py:922-939): (B,T,Ch) -> (B*T, Ch*12), channel-major."""
    x = action_ids.astype(np.float32)
    feats = []
    for k in (1.0, 2.0, 4.0, 6.0, 8.0, 10.0):
        feats += [np.cos(np.float32(k) * x), np.sin(np.float32(k) * x)]
    f = np.stack(feats, axis=-1)
    return f.reshape(x.shape[0] * x.shape[1], x.shape[2] * 12)


def conv_k_cmajor(w: torch.Tensor, taps: int) -> torch.Tensor:
    """[N, taps * C] (k = tap * C + c) -> the same weights with K in CHANNEL-BLOCK-major order, k = ((c / 64) * taps + tap) * 64
    + c % 64 (WIW_K_CMAJOR, include/wiw_svd.h): the taps of one 64-channel block become consecutive K tiles of the implicit GEMM."""
    N, K = w.shape
    C = K // taps
    assert K == taps * C and C % 64 == 0
    return w.reshape(N, taps, C // 64, 64).permute(0, 2, 1, 3).reshape(N, K).contiguous()


def conv_k_halo32(w: torch.Tensor) -> torch.Tensor:
    """[N, 9 * C] (k = tap * C + c) -> K in 32-channel blocks, k = ((c / 32) * 9 + tap) * 32 + c % 32 (WIW_K_HALO32): the K
    order of the halo-staged 3x3 convolution kernel, whose K tile is two taps of one 32-channel block."""
    N, K = w.shape
    C = K // 9
    assert K == 9 * C and C % 64 == 0
    return w.reshape(N, 9, C // 32, 32).permute(0, 2, 1, 3).reshape(N, K).contiguous()


def pack_geglu(w: torch.Tensor, b: torch.Tensor, tile: int = GEGLU_TILE):
    """Re-order the GEGLU projection [8C, K] (value rows then gate rows, activations.py:122) into tiles of
    2 * tile rows = [tile value | tile gate] so the GEMM epilogue finds both halves in one block tile (tile = 80:
    gemm.hip's 160-column tile; tile = 64: the hidden chunks of the fused FeedForward kernel, ffn.hip)."""
    GEGLU_TILE = tile  # noqa: N806  (shadows the module constant for the body below)
    n_half = w.shape[0] // 2
    pad = (-n_half) % GEGLU_TILE
    wv, wg = w[:n_half], w[n_half:]
    bv, bg = b[:n_half], b[n_half:]
    if pad:
        wv = torch.cat([wv, wv.new_zeros(pad, w.shape[1])])
        wg = torch.cat([wg, wg.new_zeros(pad, w.shape[1])])
        bv = torch.cat([bv, bv.new_zeros(pad)])
        bg = torch.cat([bg, bg.new_zeros(pad)])
    nt = (n_half + pad) // GEGLU_TILE
    wp = torch.stack([wv.reshape(nt, GEGLU_TILE, -1), wg.reshape(nt, GEGLU_TILE, -1)], dim=1).reshape(2 * (n_half + pad), -1)
    bp = torch.stack([bv.reshape(nt, GEGLU_TILE), bg.reshape(nt, GEGLU_TILE)], dim=1).reshape(-1)
    return wp.contiguous(), bp.contiguous(), n_half


def pack_ffn32(w0: torch.Tensor, b0: torch.Tensor, w2: torch.Tensor, dtype: torch.dtype):
    """Operands of `wiw_ffn32_geglu` (csrc/ffn32.hip) from the GEGLU projection [2560, 320] / [2560] (value rows then gate rows,
    activations.py:122) and net.2's weight [320, 1280], all fp32: W1 in chunks of [32 value | 32 gate] rows, b1 likewise with
    its VALUE half pre-multiplied by 0.5 (the kernel evaluates (0.5 v + 0.5 b_v) (|g| erf|g| + g)); both matrices in the tiled
    layout with the 16-row chunk swizzle.  -> (TiledW W1, b1 fp32, TiledW W2)."""
    w1p, b1p, n_half = pack_geglu(w0, b0, 32)
    b1p = b1p.clone().reshape(-1, 2, 32)
    b1p[:, 0] *= 0.5
    return (TiledW(w1p.to(dtype).contiguous(), sw16=True), b1p.reshape(-1).float().contiguous(),
            TiledW(w2.to(dtype).contiguous(), sw16=True))


def pack_temporal_qkv(wq: torch.Tensor, wk: torch.Tensor, wv: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
                      tiled: bool = True, dtype: torch.dtype = torch.bfloat16):
    """Operands of `wiw_temporal_attn_block_bf16` (temporal.hip): rows of head h = [q_h | k_h | v_h] with the LayerNorm
    weight folded in (W' = bf16(W * gamma)), and per head the fold vectors s = sum_k W'[n][k] (of the ROUNDED weights,
    so the fold is exact for what the MFMAs multiply), t = sum_k W[n][k] * beta[k]:
        LayerNorm(x) . W_n = rstd * (x . W'_n - mean * s_n) + t_n."""
    C = wq.shape[1]
    heads = C // 64
    wp = torch.stack([m.float().reshape(heads, 64, C) for m in (wq, wk, wv)], dim=1).reshape(heads * 192, C)
    wg = (wp * gamma.float()[None, :]).to(dtype).contiguous()
    fold = torch.zeros(heads, 512, dtype=torch.float32, device=wp.device)
    fold[:, :192] = wg.float().sum(dim=1).reshape(heads, 192)
    fold[:, 192:384] = (wp @ beta.float()).reshape(heads, 192)
    # the kernel streams the weight in the tiled layout of hip.TiledW (1-KiB blocks, one contiguous KiB per DMA instruction)
    return (tile_weight(wg) if tiled else wg), fold.contiguous()


def fold_layernorm(W: torch.Tensor, bias, gamma: torch.Tensor, beta: torch.Tensor, dtype: torch.dtype):
    """Operands of a GEMM with WIW_EPI_LNFOLD (include/wiw_svd.h): LayerNorm(x; gamma, beta) . W^T + bias as a GEMM on the
    RAW x.  Returns (W' = W * gamma rounded to `dtype`, lnfold = [s | t] fp32) with s_n = sum_k W'[n][k] of the ROUNDED
    rows (so the fold is exact for what the MFMAs multiply) and t_n = sum_k W[n][k] * beta[k] + bias_n."""
    Wf = W.float()
    wg = (Wf * gamma.float()[None, :]).to(dtype).contiguous()
    t = Wf @ beta.float()
    if bias is not None:
        t = t + bias.float()
    return wg, torch.stack([wg.float().sum(dim=1), t]).contiguous()


@dataclass
class RequestCond:
    """Step-invariant conditioning of one request (everything but time_embedding(t))."""
    B: int
    Bc: int
    act_emb: torch.Tensor    # fp32 [B*T, E]
    noise_emb: torch.Tensor  # fp32 [Bc, E]
    ehs_bf16: torch.Tensor   # bf16 [Bc, Dctx]
    cross: Dict[str, torch.Tensor]   # attn2 prefix -> fp32 [Bc, C]
    pos_emb: Dict[str, torch.Tensor]  # transformer prefix -> fp32 [Bc*T, C]
    pos_emb_blend: Dict[str, torch.Tensor]  # -am/(1-am) * pos_emb (AlphaBlender correction, see _transformer)
    P: int = 1                                # conditioning tokens per candidate (--num_past_obs)
    cross_kv: Optional[Dict[str, tuple]] = None   # P > 1: attn2 prefix -> (K, V) 16-bit [Bc, P, C] (to_k / to_v of the tokens)


def _require_single_key(ehs: torch.Tensor, max_tokens: int = 1) -> None:
    """The served cross-attention is the CLOSED FORM for one key (softmax over a single key is 1: the output is
    to_out(to_v(embedding)), attention.py:545-551, 740-743 with the served conditioning of one CLIP embedding per
    candidate).  A checkpoint trained with --num_past_obs > 1 hands (B, P > 1, 1024) embeddings
    (pipeline_stable_video_diffusion.py:501-504; train_svd.py:359, 889-894): those need the general attention, which a
    UNetHIP built with `num_past_obs = P` carries (round 6, csrc/cross_attn.hip, P <= 8) — anything else is refused instead
    of flattening P embeddings into one mis-shaped vector."""
    if ehs.dim() != 3 or ehs.shape[1] < 1 or ehs.shape[1] > max_tokens:
        raise NotImplementedError(f"encoder_hidden_states {tuple(ehs.shape)}: this UNetHIP takes (B, P, D) conditioning with "
                                  f"1 <= P <= {max_tokens} (built with num_past_obs = {max_tokens}; the general Sk > 1 "
                                  f"cross-attention needs UNetHIP(..., num_past_obs = P), P <= 8)")


class UNetHIP:
    def __init__(self, cfg: UNetConfig, state_dict: Dict[str, "torch.Tensor | np.ndarray"], device="cuda:0",
                 hip: Optional[Hip] = None, dtype: torch.dtype = torch.bfloat16, fold_layernorm: Optional[bool] = None,
                 residual_fp32: "Optional[bool | str]" = None, num_past_obs: int = 1):
        """dtype: 16-bit storage type of weights and activations (bf16, or fp16 = the reference's served default,
        eval_inference.py:294); with `hip` given, its dtype is used.
        residual_fp32: keep the RESIDUAL STREAM in fp32 (ABI 11), every MFMA operand stays 16-bit.  The reference keeps latents and
        the Euler update in fp32 (scheduling_euler_discrete.py:635,673) and autocasts nothing inside the UNet.
          True    (round 6) the BLOCK-LEVEL stream: the outputs x + f(x) of every ResnetBlock, the skip tensors, what enters and
                  leaves a transformer block (proj_out + x), the down / upsampler outputs are fp32 tensors, never rounded to the
                  16-bit type; a transformer block's hidden stream between proj_in and proj_out stays 16-bit and runs on the fused
                  16-bit kernels.  With fp16 storage: the reference pipeline's 25-step latents at 576x1024x14 within 5.5e-4 rms /
                  8.6e-4 max (north_star: 1e-3 relative latent error) at 0.96 x the speed of plain fp16 (profiles/r20b_*).
          "full"  (rounds 4-5) also every sub-block output inside the transformer blocks: 5.1e-4 / 8.6e-4 on the same trajectory,
                  9.1e-4 instead of 1.00e-3 on ONE forward at sigma = 15.6 — at 0.90 x the speed of fp16.
        Default: env WIW_RES32 (A/B knobs: WIW_RES32_PARTS = rb | tr | both, WIW_RES32_TR_MASK).
        num_past_obs: conditioning tokens per candidate the replica is built for (--num_past_obs, train_svd.py:359).  1 (every
        launcher of the reference): the two cross-attentions of a layer in closed form.  2..8: also the weights of the general
        form (attn2.to_q / to_k, norm2) and `wiw_cross_attn_fewkeys_bf16`; requests may then carry 1..num_past_obs tokens.
        fold_layernorm: fold norm1 / norm3 / norm_in into their consumer GEMMs at the widths the 256x160 tile serves
        (WIW_EPI_LNFOLD).  OFF by default: measured -0.7 % on the rollout (the in-kernel row statistics cost the K = 320
        GEMMs more than the four LayerNorm passes per block they replace); env WIW_LN_FOLD=1 turns it on."""
        self.cfg = cfg
        if not 1 <= int(num_past_obs) <= 8:
            raise ValueError(f"num_past_obs = {num_past_obs}: 1 .. 8 conditioning tokens per candidate")
        self.max_ctx = int(num_past_obs)
        self.device = torch.device(device)
        self.hip = hip or Hip(self.device, dtype)
        self.dtype = self.hip.dtype
        validate_state_dict(cfg, state_dict)
        self.w: Dict[str, torch.Tensor] = {}
        self.alpha: Dict[str, float] = {}
        self.no_splitk = bool(os.environ.get("WIW_NO_SPLITK"))     # A/B knob
        self.legacy_cross_guard = bool(os.environ.get("WIW_LN_ADDVEC"))
        self.ln_fold = bool(os.environ.get("WIW_LN_FOLD")) if fold_layernorm is None else bool(fold_layernorm)
        self.temporal_unfused = bool(os.environ.get("WIW_TEMPORAL_UNFUSED"))   # A/B knob: LayerNorm + QKV GEMM + attention
        # fused FeedForward kernel of the 320-channel level (ffn.hip); A/B knobs: WIW_FF_UNFUSED=1 -> two GEMMs again,
        # WIW_FFN_NO_LN=1 -> fused FeedForward behind a separate LayerNorm pass
        self.res32 = bool(os.environ.get("WIW_RES32")) if residual_fp32 is None else bool(residual_fp32)
        # WHICH part of the stream is fp32 (round 6, profiles/r20b_res32_parts.txt: the reference's 25-step trajectory at 576x1024x14,
        # fp16 weights — relative latent error rms / max and frames/s on one box, plain fp16 = 6.75e-4 / 1.09e-3, 5.857):
        #   "rb"   the BLOCK-LEVEL stream only — ResBlock outputs, skip tensors, what enters and leaves a transformer block:
        #          5.47e-4 / 8.58e-4, 5.613 frames/s = 0.958 x fp16                                  <- the mode (default)
        #   "tr"   only the hidden stream INSIDE a transformer block (proj_in .. proj_out): 1.02e-3 max, 5.424
        #   "both" rounds 4-5: 5.12e-4 / 8.56e-4, 5.242 = 0.895 x fp16                               (A/B: WIW_RES32_PARTS=both)
        # The error that accumulates is the rounding of the stream that runs through the WHOLE depth of the network; a transformer
        # block's inner stream starts fresh at proj_in and is folded back into the fp32 block stream at proj_out after 7 adds.
        # `self.res32` is the flag of the code that is running: block level, or a transformer's inside.
        assert residual_fp32 in (None, False, True, "full"), residual_fp32
        parts = os.environ.get("WIW_RES32_PARTS", "both" if residual_fp32 == "full" else "rb") if self.res32 else "none"
        assert parts in ("both", "tr", "rb", "none"), parts
        self.res32_tr, self.res32_rb = parts in ("both", "tr"), parts in ("both", "rb")
        # finer A/B knob: WHICH hidden-stream tensors of a transformer block are fp32 — bit 0 proj_in's output, 1 attn1.to_out
        # (spatial), 2 the spatial FeedForward's, 3 ff_in's, 4 attn1.to_out (temporal); "tr" / "both" = 31
        self.tr_mask = int(os.environ.get("WIW_RES32_TR_MASK", "31" if self.res32_tr else "0")) if self.res32 else 0
        self.res32_tr = self.tr_mask != 0
        self.sdt = torch.float32 if self.res32_rb else self.dtype      # dtype of block-level residual-stream tensors
        if self.res32:
            self.ln_fold = False          # the folds read the raw stream as a 16-bit MFMA operand
            # round 6: the fused temporal block runs in this mode too — on the 16-bit ROUNDING of the fp32 stream tensor, which
            # its producer writes beside it (wiw_ffn_geglu_f32stream2's out16 at C = 320, a cast pass elsewhere).  A/B knob:
            # WIW_TEMPORAL_UNFUSED_RES32=1 = LayerNorm (fp32 in) + QKV GEMM + attention core, as rounds 4-5
            self.temporal_unfused = self.temporal_unfused or bool(os.environ.get("WIW_TEMPORAL_UNFUSED_RES32"))
        # convolution weights with K in channel-block-major order (round 4: the taps of a 64-channel block re-read the same
        # activation window while it is still in L2; A/B knob WIW_K_TAPMAJOR=1 keeps the tap-major order)
        self.kc = 0 if os.environ.get("WIW_K_TAPMAJOR") else K_CMAJOR
        self.halo = not os.environ.get("WIW_CONV_NO_HALO")      # A/B knob: 3x3 convolutions without the halo-staged kernel
        self.halo_sc = not os.environ.get("WIW_CONV_NO_HALO_SC")  # A/B knob: ... except those with the fused shortcut segment
        # (fp32 stream: the fused kernel's F32E epilogue takes the fp32 residual / output; round 6, ABI 16: its in-kernel
        # LayerNorm reads the fp32 x itself — A/B knob WIW_FFN_NO_LN_RES32=1 keeps the f32in LayerNorm pass in front)
        self.ffn_fused = not os.environ.get("WIW_FF_UNFUSED") and not (self.res32 and os.environ.get("WIW_FF_UNFUSED_RES32"))
        self.ffn_ln = (self.ffn_fused and not os.environ.get("WIW_FFN_NO_LN")
                       and not (self.res32 and os.environ.get("WIW_FFN_NO_LN_RES32")))
        # round 5: the 32x32x16-MFMA form of the fused FeedForward (csrc/ffn32.hip) — correct and bit-repeatable, but NOT faster
        # than ffn.hip at M = 258 048 (805 vs 804 us: both sit on the same serial sum of MFMA + GEGLU + weight DMA + LDS time,
        # profiles/r13b_*): opt-in A/B knob, ffn.hip stays the served kernel
        self.ffn32 = self.ffn_fused and bool(os.environ.get("WIW_FFN32"))
        self._prepare(state_dict)
        self.res32_any = self.res32
        self.res32 = self.res32_rb        # from here on: the flag of the running code (see WIW_RES32_PARTS above)

    # ------------------------------------------------------------------------------------------
    # weight re-layout (once, at load)
    # ------------------------------------------------------------------------------------------
    def _t(self, sd, name) -> torch.Tensor:
        v = sd[name]
        if isinstance(v, np.ndarray):
            v = torch.from_numpy(v)
        return v.to(self.device, torch.float32)

    def _prepare(self, sd):
        cfg, w = self.cfg, self.w
        bf = self.dtype
        temb_w, temb_b = [], []
        self.temb_off: Dict[str, int] = {}
        off = 0

        def lin(p, bias=True):
            w[p + ".weight"] = self._t(sd, p + ".weight").to(bf).contiguous()
            if bias:
                w[p + ".bias"] = self._t(sd, p + ".bias").contiguous()

        def norm(p):
            w[p + ".weight"] = self._t(sd, p + ".weight").contiguous()
            w[p + ".bias"] = self._t(sd, p + ".bias").contiguous()

        n_lv = len(cfg.block_out_channels)
        # widest latent this replica is built for (default: the served 1024-px panorama): the halo-staged kernel takes rows of
        # 32 / 64 / 128 pixels (hip.conv_halo_ok), so a layer whose level can never be that wide gets no second weight copy —
        # the 1280-channel level and the mid block at the served size: ~0.5 GB of HBM per replica
        max_w = int(os.environ.get("WIW_MAX_LATENT_WIDTH", "128"))

        def level_of(p):
            if p.startswith("down_blocks."):
                return int(p.split(".")[1])
            if p.startswith("up_blocks."):
                i = int(p.split(".")[1])
                return n_lv - 1 - i - (1 if ".upsamplers." in p else 0)     # the upsampler's convolution runs at its OUTPUT size
            return n_lv - 1 if p.startswith("mid_block.") else 0

        def halo_level(p):
            return (max_w >> level_of(p)) >= 32

        def conv3(p, cin_pad=0, halo=True):
            x = self._t(sd, p + ".weight").permute(0, 2, 3, 1)  # OIHW -> OHWI
            if cin_pad and x.shape[-1] < cin_pad:
                x = torch.cat([x, x.new_zeros(*x.shape[:-1], cin_pad - x.shape[-1])], dim=-1)
            x = x.reshape(x.shape[0], -1)
            w[p + ".weight"] = (conv_k_cmajor(x, 9) if self.kc else x).to(bf).contiguous()
            w[p + ".bias"] = self._t(sd, p + ".bias").contiguous()
            # a second copy in the K order of the halo-staged kernel (N % 320 == 0: every ResnetBlock2D convolution of the
            # served widths): `_conv3` picks it when the request's geometry fits (hip.conv_halo_ok), else the copy above
            if halo and self.halo and halo_level(p) and x.shape[0] % 320 == 0 and x.shape[1] % (9 * 64) == 0:
                w[p + ".weight_h"] = conv_k_halo32(x).to(bf).contiguous()

        def convt(p):
            x = self._t(sd, p + ".weight")[:, :, :, 0, 0].permute(0, 2, 1)  # (O,I,3) -> (O,3,I)
            x = x.reshape(x.shape[0], -1)
            w[p + ".weight"] = (conv_k_cmajor(x, 3) if self.kc else x).to(bf).contiguous()
            w[p + ".bias"] = self._t(sd, p + ".bias").contiguous()

        def res(p):
            nonlocal off
            s, t = p + ".spatial_res_block", p + ".temporal_res_block"
            norm(s + ".norm1"); conv3(s + ".conv1"); norm(s + ".norm2"); conv3(s + ".conv2")
            if s + ".conv_shortcut.weight" in sd and os.environ.get("WIW_UNFUSED_SHORTCUT"):   # A/B knob
                x = self._t(sd, s + ".conv_shortcut.weight")[:, :, 0, 0]
                w[s + ".conv_shortcut.weight"] = x.to(bf).contiguous()
                w[s + ".conv_shortcut.bias"] = self._t(sd, s + ".conv_shortcut.bias").contiguous()
            elif s + ".conv_shortcut.weight" in sd:
                # the 1x1 shortcut over the (skip-concatenated) block input is a tenth K segment of conv2's implicit
                # GEMM: W = [conv2 (9*Cout) | shortcut (Cin)], bias = b2 + b_sc — the shortcut tensor is never written
                x = self._t(sd, s + ".conv_shortcut.weight")[:, :, 0, 0]
                w2 = self._t(sd, s + ".conv2.weight").permute(0, 2, 3, 1).reshape(x.shape[0], -1)
                if self.kc:
                    w2 = conv_k_cmajor(w2, 9)          # the shortcut segment stays behind the taps
                w[s + ".conv2sc.weight"] = torch.cat([w2, x], dim=1).to(bf).contiguous()
                w[s + ".conv2sc.bias"] = (self._t(sd, s + ".conv2.bias") + self._t(sd, s + ".conv_shortcut.bias")).contiguous()
                del w[s + ".conv2.weight"], w[s + ".conv2.bias"]
                w.pop(s + ".conv2.weight_h", None)
                if (self.halo and self.halo_sc and halo_level(s) and x.shape[0] % 320 == 0 and w2.shape[1] % (9 * 64) == 0
                        and x.shape[1] % 64 == 0):
                    w2h = conv_k_halo32(self._t(sd, s + ".conv2.weight").permute(0, 2, 3, 1).reshape(x.shape[0], -1))
                    w[s + ".conv2sc.weight_h"] = torch.cat([w2h, x], dim=1).to(bf).contiguous()
            norm(t + ".norm1"); convt(t + ".conv1"); norm(t + ".norm2"); convt(t + ".conv2")
            for q in (s, t):  # all time_emb_proj layers are evaluated by ONE batched GEMM per step
                tw = self._t(sd, q + ".time_emb_proj.weight")
                self.temb_off[q] = off
                off += tw.shape[0]
                temb_w.append(tw)
                temb_b.append(self._t(sd, q + ".time_emb_proj.bias"))
            self.alpha[p] = float(torch.sigmoid(self._t(sd, p + ".time_mixer.mix_factor")).item())

        def ff(p, ln=None):
            """ln: prefix of the LayerNorm that feeds this FeedForward; when its width is served by the 256x160 tile
            (C < 640: the LayerNorm fold lives in gemm.hip) a second, gamma-folded copy of the packed GEGLU projection
            and its [s | t] vectors are prepared (both packed in the same row order)."""
            w0, b0 = self._t(sd, p + ".net.0.proj.weight"), self._t(sd, p + ".net.0.proj.bias")
            wp, bp, n_half = pack_geglu(w0, b0)
            w[p + ".net.0.proj.weight"] = wp.to(bf).contiguous()
            w[p + ".net.0.proj.bias"] = bp.contiguous()
            if ln is not None and self._fold_ln(w0.shape[1]):
                gamma, beta = self._t(sd, ln + ".weight"), self._t(sd, ln + ".bias")
                wgp, tp, _ = pack_geglu(w0 * gamma[None, :], w0 @ beta + b0)
                wgp = wgp.to(bf).contiguous()
                w[p + ".net.0.proj.lnfold.weight"] = wgp
                w[p + ".net.0.proj.lnfold.st"] = torch.stack([wgp.float().sum(dim=1), tp.float()]).contiguous()
            lin(p + ".net.2")
            if self.ffn_fused and w0.shape == (2 * FFN_HIDDEN, FFN_C):
                # operands of wiw_ffn_geglu_bf16: W1 in chunks of [64 value | 64 gate] rows; with the LayerNorm fused, gamma
                # goes into W1 and beta into b1 (the kernel normalises the raw rows in registers)
                w1, b1, _ = pack_geglu(w0, b0, FFN_CHUNK)
                w[p + ".ffn.w1"], w[p + ".ffn.b1"] = TiledW(w1.to(bf).contiguous()), b1.contiguous()
                if self.ffn32:
                    w[p + ".ffn32"] = pack_ffn32(w0, b0, self._t(sd, p + ".net.2.weight"), bf)
                if ln is not None and self.ffn_ln:
                    gamma, beta = self._t(sd, ln + ".weight"), self._t(sd, ln + ".bias")
                    w1, b1, _ = pack_geglu(w0 * gamma[None, :], w0 @ beta + b0, FFN_CHUNK)
                    w[p + ".ffn.w1ln"], w[p + ".ffn.b1ln"] = TiledW(w1.to(bf).contiguous()), b1.contiguous()
                    if self.ffn32:
                        w[p + ".ffn32ln"] = pack_ffn32(w0 * gamma[None, :], w0 @ beta + b0, self._t(sd, p + ".net.2.weight"), bf)

        def transformer(p):
            norm(p + ".norm"); lin(p + ".proj_in"); lin(p + ".proj_out")
            # ATTN_PRESCALE bakes head_dim 64 into the packed to_q rows (and the kernels are built for d = 64): a checkpoint with
            # another head width must fail here, not run with a silently wrong softmax temperature
            Cp = self._t(sd, p + ".proj_in.weight").shape[0]
            lvl = [i for i, c in enumerate(cfg.block_out_channels) if c == Cp]
            assert lvl and all(Cp == 64 * cfg.num_attention_heads[i] for i in lvl), (
                f"{p}: {Cp} channels with heads {[cfg.num_attention_heads[i] for i in lvl]}: this build serves head_dim 64 only")
            b, t = p + ".transformer_blocks.0", p + ".temporal_transformer_blocks.0"
            norm(b + ".norm1"); norm(b + ".norm3"); norm(t + ".norm_in"); norm(t + ".norm1"); norm(t + ".norm3")
            # to_q rows PRE-SCALED by log2(e) / sqrt(64) in fp32, before their one rounding to 16 bits: the spatial attention
            # kernel (csrc/attention32.hip) then has the base-2 exponent straight out of Q.K (attention_processor.py:2383
            # applies 1/sqrt(d) inside F.scaled_dot_product_attention)
            wqkv = torch.cat([self._t(sd, b + ".attn1.to_q.weight") * ATTN_PRESCALE, self._t(sd, b + ".attn1.to_k.weight"),
                              self._t(sd, b + ".attn1.to_v.weight")])
            w[b + ".attn1.to_qkv.weight"] = wqkv.to(bf).contiguous()
            if self._fold_ln(wqkv.shape[1]) and wqkv.shape[0] % 160 == 0:   # norm1 folded into the q|k|v projection
                w[b + ".attn1.to_qkv.lnfold.weight"], w[b + ".attn1.to_qkv.lnfold.st"] = fold_layernorm(
                    wqkv, None, self._t(sd, b + ".norm1.weight"), self._t(sd, b + ".norm1.bias"), bf)
            lin(b + ".attn1.to_out.0")
            w[t + ".attn1.to_qkv.weight"] = torch.cat([self._t(sd, t + ".attn1.to_q.weight"),
                                                       self._t(sd, t + ".attn1.to_k.weight"),
                                                       self._t(sd, t + ".attn1.to_v.weight")]).to(bf).contiguous()
            lin(t + ".attn1.to_out.0")
            w[t + ".attn1.fused.weight"], w[t + ".attn1.fused.fold"] = pack_temporal_qkv(
                self._t(sd, t + ".attn1.to_q.weight"), self._t(sd, t + ".attn1.to_k.weight"),
                self._t(sd, t + ".attn1.to_v.weight"), self._t(sd, t + ".norm1.weight"), self._t(sd, t + ".norm1.bias"),
                dtype=bf)
            for q in (b, t):  # single-key cross-attention: only to_v and to_out matter (§9.3)
                lin(q + ".attn2.to_v", bias=False); lin(q + ".attn2.to_out.0")
                if self.max_ctx > 1:   # the general form: Q projection of the normalised stream, K projection of the tokens
                    norm(q + ".norm2"); lin(q + ".attn2.to_q", bias=False); lin(q + ".attn2.to_k", bias=False)
            ff(b + ".ff", b + ".norm3"); ff(t + ".ff_in", t + ".norm_in"); ff(t + ".ff", t + ".norm3")
            lin(p + ".time_pos_embed.linear_1"); lin(p + ".time_pos_embed.linear_2")
            self.alpha[p] = float(torch.sigmoid(self._t(sd, p + ".time_mixer.mix_factor")).item())

        conv3("conv_in", cin_pad=CIN_PAD)
        conv3("conv_out")
        norm("conv_norm_out")
        for p in ("time_embedding", "add_embedding_action", "add_embedding_noise"):
            lin(p + ".linear_1"); lin(p + ".linear_2")
        # action projection: K = 12 * channels padded to a multiple of 64
        aw = self._t(sd, "add_action_proj.proj.weight")
        self.act_kpad = (-aw.shape[1]) % 64
        w["add_action_proj.proj.weight"] = torch.cat([aw, aw.new_zeros(aw.shape[0], self.act_kpad)], dim=1).to(bf).contiguous()
        w["add_action_proj.proj.bias"] = self._t(sd, "add_action_proj.proj.bias").contiguous()
        n = len(cfg.block_out_channels)
        self.res_names: List[str] = []
        self.tr_names: List[str] = []
        for i in range(n):
            for j in range(cfg.layers_per_block):
                self.res_names.append(f"down_blocks.{i}.resnets.{j}")
                if i < n - 1:
                    self.tr_names.append(f"down_blocks.{i}.attentions.{j}")
            if i < n - 1:
                conv3(f"down_blocks.{i}.downsamplers.0.conv", halo=False)   # stride 2: never reaches `_conv3` (no second copy)
        self.res_names += ["mid_block.resnets.0", "mid_block.resnets.1"]
        self.tr_names.append("mid_block.attentions.0")
        for i in range(n):
            for j in range(cfg.layers_per_block + 1):
                self.res_names.append(f"up_blocks.{i}.resnets.{j}")
                if i > 0:
                    self.tr_names.append(f"up_blocks.{i}.attentions.{j}")
            if i < n - 1:
                conv3(f"up_blocks.{i}.upsamplers.0.conv")
        for p in self.res_names:
            res(p)
        for p in self.tr_names:
            transformer(p)
        w["temb_all.weight"] = torch.cat(temb_w).to(bf).contiguous()
        w["temb_all.bias"] = torch.cat(temb_b).contiguous()
        self.temb_total = off
        # Every static GEMM weight is re-laid-out for the LDS-DMA stream (hip.TiledW).  Exceptions: `attn1.to_v` of the
        # spatial block is the A operand of a swapped-operand GEMM (V^T = Wv . a^T) and stays row-major; the fused
        # temporal weights are tiled by pack_temporal_qkv.
        if not os.environ.get("WIW_W_UNTILED"):    # A/B knob
            for k in list(w):
                t_ = w[k]
                if ((k.endswith(".weight") or k.endswith(".weight_h")) and torch.is_tensor(t_) and t_.dim() == 2 and t_.dtype == bf
                        and t_.shape[1] % 64 == 0 and not k.endswith(".attn1.to_v.weight") and ".fused." not in k):
                    w[k] = TiledW(t_)
        for k in list(w):   # the fused FeedForward kernel streams W2 tiled whatever the A/B knob above says
            if k.endswith(".ffn.w1") and torch.is_tensor(w[k[:-len("ffn.w1")] + "net.2.weight"]):
                w[k[:-len("ffn.w1")] + "net.2.weight"] = TiledW(w[k[:-len("ffn.w1")] + "net.2.weight"])
        # frame-position embeddings depend only on the frame index -> computed once (transformer_temporal.py:329-339)
        T = cfg.num_frames
        self.pos_emb_T: Dict[str, torch.Tensor] = {}
        for p in self.tr_names:
            Cn = w[p + ".proj_in.weight"].shape[0]
            feat = torch.from_numpy(sinusoid(np.arange(T), Cn)).to(self.device, bf)
            self.pos_emb_T[p] = self._mlp(p + ".time_pos_embed", feat, T)
        torch.cuda.synchronize(self.device)

    # ------------------------------------------------------------------------------------------
    # small helpers
    # ------------------------------------------------------------------------------------------
    def _empty(self, *shape, dtype=None):
        return torch.empty(shape, dtype=dtype or self.dtype, device=self.device)

    def _fold_ln(self, C: int) -> bool:
        """LayerNorm folded into its consumer GEMM (WIW_EPI_LNFOLD) where that GEMM runs on the 256x160 / 128x160 tile,
        i.e. K = C < 640 (the C = 320 level of the served network): decided by the layer width, never by the batch."""
        return C < 640 and self.ln_fold

    def _splitk(self, rows_per_item, N, K):
        """Split-K factor of an implicit-GEMM conv.  At the innermost 1280-channel level one candidate has M = 4032 rows:
        16 x 4 = 64 tiles of 256 x 320 for 256 CUs.  Cutting K into 4 ranges (256 items, fp32 partial slabs, a reduce
        kernel that applies the epilogue) measured -17 % at K = 11 520 and -31 % at K = 23 040; at K <= 5120 (temporal
        convs, FF down-projections) the slab traffic costs more than the idle CUs (+30...+45 %), so those stay whole.
        The factor is decided from the rows of ONE candidate (2 CFG items), never from the batch in flight: a candidate is
        evaluated with the same arithmetic alone and inside any batch (the bit-exact batch contract)."""
        if self.no_splitk or N % 320 or K < 8192:
            return 1
        tiles, nk = -(-2 * rows_per_item // 256) * (N // 320), K // 64
        if tiles > 100:
            return 1
        for sk in (4, 2, 3):
            if nk % sk == 0 and tiles * sk >= 200:
                return sk
        return 1

    def _linear(self, x, p, M, *, out_f32=False, silu=False, res1=None, stream=False, **kw):
        """stream=True: the output is a residual-stream tensor (fp32 when `residual_fp32`)."""
        W = self.w[p + ".weight"]
        N, K = W.shape
        out_f32 = out_f32 or (stream and self.res32)
        out = self._empty(M, N, dtype=torch.float32 if out_f32 else self.dtype)
        epi = (EPI_OUT_F32 if out_f32 else 0) | (EPI_SILU if silu else 0)
        return self.hip.gemm(x, W, out, M=M, N=N, K=K, C1=K, bias=self.w.get(p + ".bias"), epilogue=epi,
                             res1=res1, ldr1=N if res1 is not None else 0, beta1=1.0 if res1 is not None else 0.0, **kw)

    def _mlp(self, p, x_bf16, M) -> torch.Tensor:
        """TimestepEmbedding (embeddings.py:804-816): linear_1 -> SiLU -> linear_2; fp32 result [M, out]."""
        h = self._linear(x_bf16, p + ".linear_1", M, silu=True)
        return self._linear(h, p + ".linear_2", M, out_f32=True)

    def _geglu_ff(self, a, p, M, Cn, ln_input=None, stream=True, want16=False, **epi_kw):
        """FeedForward with GEGLU (attention.py:1185-1243): returns GEMM2 output with the given epilogue.
        ln_input: the RAW input of the LayerNorm in front of this FeedForward — given instead of `a` (= None) where the
        norm is folded into the projection (`_fold_ln`).  stream: the output is a residual-stream tensor (fp32 when
        `residual_fp32`); False for the AlphaBlender output that only feeds proj_out as an MFMA operand.
        want16 (fp32 stream only): also return the output's 16-bit rounding -> (out, out16); written by the fused kernel's
        epilogue at C = 320 (ABI 16), by a cast pass behind the second GEMM elsewhere."""
        if want16:
            assert self.res32 and stream
        if (p + ".ffn.w1") in self.w:     # C = 320: ONE kernel, the [M, 4C] hidden tensor never exists (ffn.hip)
            ln = ln_input is not None
            assert not ln or (p + ".ffn.w1ln") in self.w
            out = self._empty(M, Cn, dtype=torch.float32 if (stream and self.res32) else self.dtype)
            kw = {k: v for k, v in epi_kw.items() if k in ("rowvec", "rowvec_ld", "rows_per_vec", "res1", "ldr1", "beta1",
                                                           "res2", "ldr2", "beta2", "alpha")}
            assert len(kw) == len(epi_kw), f"unsupported FeedForward epilogue arguments: {set(epi_kw) - set(kw)}"
            if (p + ".ffn32") in self.w and not self.res32:      # opt-in (WIW_FFN32=1): the 32x32x16 form
                W1s, b1s, W2s = self.w[p + (".ffn32ln" if ln else ".ffn32")]
                return self.hip.ffn32_geglu(ln_input if ln else a, W1s, b1s, W2s, self.w[p + ".net.2.bias"], out, M, ln=ln,
                                            ln_eps=1e-5, **kw)
            o16 = self._empty(M, Cn) if want16 else None
            self.hip.ffn_geglu(ln_input if ln else a, self.w[p + (".ffn.w1ln" if ln else ".ffn.w1")],
                               self.w[p + (".ffn.b1ln" if ln else ".ffn.b1")], self.w[p + ".net.2.weight"],
                               self.w[p + ".net.2.bias"], out, M, ln=ln, ln_eps=1e-5, out16=o16, **kw)
            return (out, o16) if want16 else out
        g = self._empty(M, 4 * Cn)
        if ln_input is not None:
            W1 = self.w[p + ".net.0.proj.lnfold.weight"]
            self.hip.gemm(ln_input, W1, g, M=M, N=W1.shape[0], K=Cn, C1=Cn, epilogue=EPI_GEGLU, n_out=4 * Cn,
                          lnfold=self.w[p + ".net.0.proj.lnfold.st"], ln_eps=1e-5)
        else:
            W1 = self.w[p + ".net.0.proj.weight"]
            self.hip.gemm(a, W1, g, M=M, N=W1.shape[0], K=Cn, C1=Cn, bias=self.w[p + ".net.0.proj.bias"],
                          epilogue=EPI_GEGLU, n_out=4 * Cn)
        W2 = self.w[p + ".net.2.weight"]
        f32 = stream and self.res32
        out = self._empty(M, Cn, dtype=torch.float32 if f32 else self.dtype)
        self.hip.gemm(g, W2, out, M=M, N=Cn, K=4 * Cn, C1=4 * Cn, bias=self.w[p + ".net.2.bias"],
                      epilogue=EPI_OUT_F32 if f32 else 0, **epi_kw)
        return (out, self.hip.cast16(out)) if want16 else out

    # ------------------------------------------------------------------------------------------
    # request-level (step-invariant) conditioning
    # ------------------------------------------------------------------------------------------
    def prepare_request(self, image_embeddings: torch.Tensor, action_ids: np.ndarray, noise_aug_strength: float = 0.02,
                        cfg_batch: bool = True) -> RequestCond:
        """image_embeddings: (B,1,Dctx) CLIP embeds of the cond halves (uncond halves are zeros,
        pipeline:221-227); action_ids: (B,T,Ch) from `action_ids_idx_encode`."""
        cfg, w = self.cfg, self.w
        B = image_embeddings.shape[0]
        Bc = 2 * B if cfg_batch else B
        T = cfg.num_frames
        bf = self.dtype
        _require_single_key(image_embeddings, self.max_ctx)
        P = image_embeddings.shape[1]
        ie = image_embeddings.reshape(B * P, -1).to(self.device, torch.float32)
        ehs = torch.cat([torch.zeros_like(ie), ie]) if cfg_batch else ie      # [Bc * P, D]: item-major, the P tokens of an item adjacent
        ehs = ehs.to(bf).contiguous()
        # actions: Fourier features -> proj -> MLP (unet:472, 274-280)
        feat = action_features(np.asarray(action_ids))
        if self.act_kpad:
            feat = np.concatenate([feat, np.zeros((feat.shape[0], self.act_kpad), np.float32)], axis=1)
        feat = torch.from_numpy(feat).to(self.device, bf)
        proj = self._linear(feat, "add_action_proj.proj", B * T)
        act = self._mlp("add_embedding_action", proj, B * T)
        # noise-aug embedding (unet:484-486); same value for every CFG row
        nfeat = torch.from_numpy(sinusoid(np.full((Bc,), noise_aug_strength, np.float32), cfg.addition_time_embed_dim))
        noise = self._mlp("add_embedding_noise", nfeat.to(self.device, bf), Bc)
        cross, pos, posb, cross_kv = {}, {}, {}, {}
        for p in self.tr_names:
            for q in (p + ".transformer_blocks.0", p + ".temporal_transformer_blocks.0"):
                v = self._linear(ehs, q + ".attn2.to_v", Bc * P)
                if P == 1:      # softmax over one key == 1: attn2(x) = to_out(to_v(token)), one vector per CFG item
                    cross[q] = self._linear(v, q + ".attn2.to_out.0", Bc, out_f32=True)
                else:           # K / V of the item's tokens, [Bc, P, C]: the per-step part is Q.K softmax V (csrc/cross_attn.hip)
                    k = self._linear(ehs, q + ".attn2.to_k", Bc * P)
                    cross_kv[q] = (k.reshape(Bc, P, -1), v.reshape(Bc, P, -1))
            pos[p] = self.pos_emb_T[p].repeat(Bc, 1).contiguous()
            am = self.alpha[p]
            posb[p] = (pos[p] * (-am / (1.0 - am))).contiguous() if abs(1.0 - am) > 1e-4 else pos[p]
        return RequestCond(B=B, Bc=Bc, act_emb=act, noise_emb=noise, ehs_bf16=ehs, cross=cross, pos_emb=pos,
                           pos_emb_blend=posb, P=P, cross_kv=cross_kv if P > 1 else None)

    # ------------------------------------------------------------------------------------------
    # blocks
    # ------------------------------------------------------------------------------------------
    def _res_block(self, p, x1, C1, x2, C2, Cout, M, H, W, temb_all, eps):
        """SpatioTemporalResBlock (resnet.py:686-716) on token-major tensors; (x1|x2) is the channel concat."""
        hip, w, T = self.hip, self.w, self.cfg.num_frames
        S = H * W
        s, t = p + ".spatial_res_block", p + ".temporal_res_block"
        Cin = C1 + C2
        sdt = self.sdt                       # residual-stream dtype: xs and the block output (fp32 when residual_fp32)
        epi_s = EPI_OUT_F32 if self.res32 else 0
        raw = None
        if self.res32 and s + ".conv2sc.weight" in w:
            # the fused 1x1 shortcut reads the block input as an MFMA operand: the GroupNorm pass (which holds it in
            # registers anyway) also writes its rounded copy, already concatenated
            raw = self._empty(M, Cin)
        xn = hip.groupnorm(x1, C1, x2, C2, M, S, w[s + ".norm1.weight"], w[s + ".norm1.bias"], eps, True, raw16=raw)
        h = self._empty(M, Cout)
        self._conv3(xn, s + ".conv1", h, M=M, N=Cout, C=Cin, H=H, W=W, splitk=self._splitk(T * S, Cout, 9 * Cin),
                    rowvec=temb_all[:, self.temb_off[s]:], rowvec_ld=self.temb_total, rows_per_vec=S)
        hn = hip.groupnorm(h, Cout, None, 0, M, S, w[s + ".norm2.weight"], w[s + ".norm2.bias"], eps, True)
        xs = self._empty(M, Cout, dtype=sdt)
        if s + ".conv2sc.weight" in w:     # conv2 + 1x1 shortcut over (x1 | x2) in one implicit GEMM
            sc_ops = dict(A2=raw, C2=Cin) if raw is not None else dict(A2=x1, C2=C1, A3=x2, C3=C2)
            self._conv3(hn, s + ".conv2sc", xs, M=M, N=Cout, C=Cout, H=H, W=W, splitk=self._splitk(T * S, Cout, 9 * Cout + Cin),
                        epilogue=epi_s, Ksc=Cin, **sc_ops)
        else:
            if s + ".conv_shortcut.weight" in w:   # unfused A/B path: separate 1x1 GEMM, then residual
                assert not self.res32, "WIW_UNFUSED_SHORTCUT is a 16-bit-stream A/B knob"
                sc = self._empty(M, Cout)
                hip.gemm(x1, w[s + ".conv_shortcut.weight"], sc, M=M, N=Cout, K=Cin, C1=C1, A2=x2, C2=C2,
                         bias=w[s + ".conv_shortcut.bias"])
            else:
                assert x2 is None
                sc = x1
            self._conv3(hn, s + ".conv2", xs, M=M, N=Cout, C=Cout, H=H, W=W, splitk=self._splitk(T * S, Cout, 9 * Cout),
                        res1=sc, ldr1=Cout, beta1=1.0, epilogue=epi_s)
        # temporal resnet: GroupNorm statistics run over (T, H, W) of each batch item (5-D input, resnet.py:611)
        xn = hip.groupnorm(xs, Cout, None, 0, M, T * S, w[t + ".norm1.weight"], w[t + ".norm1.bias"], eps, True, clip=True)
        hip.gemm(xn, w[t + ".conv1.weight"], h, M=M, N=Cout, K=3 * Cout, C1=Cout, mode=A_CONV_T3, H=H, Wd=W, T=T,
                 bias=w[t + ".conv1.bias"], rowvec=temb_all[:, self.temb_off[t]:], rowvec_ld=self.temb_total,
                 rows_per_vec=S, epilogue=self.kc)
        hn = hip.groupnorm(h, Cout, None, 0, M, T * S, w[t + ".norm2.weight"], w[t + ".norm2.bias"], eps, True, out=hn, clip=True)
        a = self.alpha[p]
        out = self._empty(M, Cout, dtype=sdt)
        # AlphaBlender: a*xs + (1-a)*(xs + conv2(h) + b) = xs + (1-a)*(acc + b)   (resnet.py:784-797)
        hip.gemm(hn, w[t + ".conv2.weight"], out, M=M, N=Cout, K=3 * Cout, C1=Cout, mode=A_CONV_T3, H=H, Wd=W, T=T,
                 bias=w[t + ".conv2.bias"], alpha=1.0 - a, res1=xs, ldr1=Cout, beta1=1.0, epilogue=epi_s | self.kc)
        return out

    def _conv3(self, x, key, out, *, M, N, C, H, W, splitk=1, epilogue=0, Ksc=0, mode=A_CONV3X3, **kw):
        """One 3x3 convolution (stride 1, pad 1; Ksc > 0: + the fused 1x1 shortcut segment over A2 | A3 in `kw`; mode
        A_CONV3X3_UP: behind a nearest x2 upsample, H x W the OUTPUT size) as an implicit GEMM: the halo-staged kernel (weight
        copy `key.weight_h`, WIW_K_HALO32) when the geometry and the epilogue fit it, else the per-tap kernels on `key.weight`."""
        w = self.w
        wh = w.get(key + ".weight_h")
        # The two kernels sum K in different orders, so WHICH one runs must not depend on the batch in flight: the geometry
        # test is made on the rows of ONE candidate as well (18 x 32 frames straddle tiles: 28 CFG frames of a candidate are
        # 63 whole tiles, the 14 frames of a candidate without CFG are not — then every batch size takes the per-tap kernel)
        up = mode == A_CONV3X3_UP
        if (wh is not None and splitk <= 1 and not (epilogue & ~EPI_OUT_F32) and isinstance(wh, TiledW)
                and self.hip.conv_halo_ok(M, N, C, H, W, up=up)
                and self.hip.conv_halo_ok(M // max(getattr(self, "_ncand", 1), 1), N, C, H, W, up=up)):
            self.hip.gemm(x, wh, out, M=M, N=N, K=9 * C + Ksc, C1=C, mode=mode, H=H, Wd=W, bias=w[key + ".bias"],
                          epilogue=K_HALO32 | epilogue, **kw)
        else:
            self.hip.gemm(x, w[key + ".weight"], out, M=M, N=N, K=9 * C + Ksc, C1=C, mode=mode, H=H, Wd=W,
                          bias=w[key + ".bias"], splitk=splitk, epilogue=epilogue | self.kc, **kw)

    def _cross_attention(self, q, h, M, Cn, rows_per_item, heads, cond: RequestCond):
        """x + attn2(norm2(x), tokens) with P > 1 conditioning tokens (attention.py:545-551 / 740-743): LayerNorm, Q projection,
        the few-keys attention against the item's (K, V) of `prepare_request`, out-projection + residual.  The rows of a CFG
        item are consecutive in this build's token order for the spatial AND the temporal block (its time_context is the
        first frame's tokens = the item's, transformer_temporal.py:313-320), so both use rows_per_item = T * S."""
        hip, w = self.hip, self.w
        if self.legacy_cross_guard:
            raise NotImplementedError("WIW_LN_ADDVEC (A/B knob of the single-key closed form) with num_past_obs > 1")
        a2 = hip.layernorm(h, M, Cn, w[q + ".norm2.weight"], w[q + ".norm2.bias"])
        q2 = self._linear(a2, q + ".attn2.to_q", M)
        K, V = cond.cross_kv[q]
        o2 = hip.cross_attn_fewkeys(q2, Cn, K, V, a2, Cn, M, rows_per_item, heads, cond.P, 1.0 / math.sqrt(64.0))
        return self._linear(o2, q + ".attn2.to_out.0", M, res1=h, stream=h.dtype == torch.float32)   # stays on h's stream

    def _transformer(self, p, x, Cn, M, H, W, heads, cond: RequestCond):
        """TransformerSpatioTemporalModel (transformer_temporal.py:279-382), one layer."""
        rb = self.res32
        self.res32 = self.res32_tr            # the hidden stream between proj_in and proj_out
        try:
            hb = self._transformer_inside(p, x, Cn, M, H, W, heads, cond)
        finally:
            self.res32 = rb
        return self._linear(hb, p + ".proj_out", M, res1=x, stream=True)      # back on the block-level stream

    def _transformer_inside(self, p, x, Cn, M, H, W, heads, cond: RequestCond):
        hip, w, T = self.hip, self.w, self.cfg.num_frames
        S = H * W
        frames = M // S
        batch = frames // T
        b, t = p + ".transformer_blocks.0", p + ".temporal_transformer_blocks.0"
        scale = 1.0 / math.sqrt(64.0)
        xn = hip.groupnorm(x, Cn, None, 0, M, S, w[p + ".norm.weight"], w[p + ".norm.bias"], 1e-6, False)
        tb = lambda k: bool((self.tr_mask >> k) & 1)      # noqa: E731  (is hidden-stream tensor k fp32?)
        h = self._linear(xn, p + ".proj_in", M, stream=tb(0))
        # ---- spatial block (attention.py:462-582)
        legacy = bool(os.environ.get("WIW_LN_ADDVEC"))   # A/B knob: the adds inside the LayerNorm kernel (extra write pass)
        # LayerNorm folded into its consumer GEMM where that GEMM runs on the 256x160 tile (`_fold_ln`): the projection
        # reads the RAW residual stream, the kernel derives mean / rstd of its rows from the operand fragments
        fold_qkv = (b + ".attn1.to_qkv.lnfold.weight") in w
        fold_ff = ((b + ".ff.net.0.proj.lnfold.weight") in w or (b + ".ff.ffn.w1ln") in w) and not legacy
        a = xn
        if self.res32:
            assert not (fold_qkv or legacy), "A/B knobs of the 16-bit stream"
        if not fold_qkv:
            a = hip.layernorm(h, M, Cn, w[b + ".norm1.weight"], w[b + ".norm1.bias"], out=xn)
        vt = self._empty(Cn, M)
        o = self._empty(M, Cn)
        # ONE q|k|v projection (the activation is read once), then a 64x64-tiled transpose of the V columns
        qkv = self._empty(M, 3 * Cn)
        if fold_qkv:
            hip.gemm(h, w[b + ".attn1.to_qkv.lnfold.weight"], qkv, M=M, N=3 * Cn, K=Cn, C1=Cn,
                     lnfold=w[b + ".attn1.to_qkv.lnfold.st"], ln_eps=1e-5)
        else:
            hip.gemm(a, w[b + ".attn1.to_qkv.weight"], qkv, M=M, N=3 * Cn, K=Cn, C1=Cn)
        hip.transpose(qkv, 3 * Cn, 2 * Cn, M, Cn, vt, M)
        hip.attn_spatial_ps(qkv, 3 * Cn, Cn, vt, M, o, Cn, frames, S, heads)    # Q arrives pre-scaled (to_qkv weights)
        # The adds that follow a GEMM in the reference — the single-key cross-attention output (one vector per CFG item,
        # attention.py:545-551, 740-743) and the frame-position embedding (transformer_temporal.py:352-353) — ride in that
        # GEMM's epilogue as its per-row-group vector, so every LayerNorm below is a plain one-read / one-write pass.
        if legacy:
            h = self._linear(o, b + ".attn1.to_out.0", M, res1=h)
            a = hip.layernorm(h, M, Cn, w[b + ".norm3.weight"], w[b + ".norm3.bias"], addvec=cond.cross[b], addvec_ld=Cn,
                              rows_per_vec=T * S, sum_out=h, out=a)
        elif cond.P > 1:
            h = self._linear(o, b + ".attn1.to_out.0", M, res1=h, stream=tb(1))
            h = self._cross_attention(b, h, M, Cn, T * S, heads, cond)
            if not fold_ff:
                a = hip.layernorm(h, M, Cn, w[b + ".norm3.weight"], w[b + ".norm3.bias"], out=a)
        else:
            h = self._linear(o, b + ".attn1.to_out.0", M, res1=h, rowvec=cond.cross[b], rowvec_ld=Cn, rows_per_vec=T * S,
                             stream=tb(1))
            if not fold_ff:
                a = hip.layernorm(h, M, Cn, w[b + ".norm3.weight"], w[b + ".norm3.bias"], out=a)
        am = self.alpha[p]
        # hs + emb is stored instead of hs; the blend below subtracts am * emb again
        fold_emb = abs(1.0 - am) > 1e-4 and not legacy
        if self.res32 and not fold_emb:
            raise NotImplementedError("residual_fp32 with sigmoid(mix_factor) == 1: the pre-add LayerNorm path is 16-bit only")
        lnin = h if fold_ff else None         # the FeedForwards below take the RAW stream where their norm is folded
        if fold_emb:
            hm = self._geglu_ff(a, b + ".ff", M, Cn, ln_input=lnin, res1=h, ldr1=Cn, beta1=1.0, rowvec=cond.pos_emb[p],
                                rowvec_ld=Cn, rows_per_vec=S, stream=tb(2))
            hs = hm                           # = spatial output + emb
            if not fold_ff:
                a = hip.layernorm(hm, M, Cn, w[t + ".norm_in.weight"], w[t + ".norm_in.bias"], out=a)
        else:
            hs = self._geglu_ff(a, b + ".ff", M, Cn, ln_input=lnin, res1=h, ldr1=Cn, beta1=1.0, stream=tb(2))
            hm = self._empty(M, Cn)
            a = hip.layernorm(hs, M, Cn, w[t + ".norm_in.weight"], w[t + ".norm_in.bias"], addvec=cond.pos_emb[p],
                              addvec_ld=Cn, rows_per_vec=S, sum_out=hm, out=a)
        # ---- temporal block (attention.py:707-762); rows stay in (b,t,s) order
        t_fused = T <= 14 and not self.temporal_unfused and M * Cn * 2 < (1 << 32)
        w16 = tb(3) and t_fused               # ff_in's output is fp32 and the fused temporal block wants its 16-bit rounding
        hm = self._geglu_ff(a, t + ".ff_in", M, Cn, ln_input=hm if (fold_ff and fold_emb) else None, res1=hm, ldr1=Cn, beta1=1.0,
                            stream=tb(3), want16=w16)
        if t_fused:
            # norm1 + to_q/k/v + the 14x14 attention in ONE kernel (temporal.hip): LayerNorm folded into the projection,
            # Q/K/V never leave the registers — no LayerNorm pass, no 3C-wide QKV tensor.  (fp32 stream: the kernel reads the
            # stream tensor's 16-bit rounding, written beside it by its producer.)
            hm, hm16 = hm if w16 else (hm, hm)
            hip.temporal_attn_block(hm16, w[t + ".attn1.fused.weight"], w[t + ".attn1.fused.fold"], o, Cn, batch, T, S,
                                    heads, 1e-5, scale)
        else:
            a = hip.layernorm(hm, M, Cn, w[t + ".norm1.weight"], w[t + ".norm1.bias"], out=a)
            qkv = self._empty(M, 3 * Cn)
            hip.gemm(a, w[t + ".attn1.to_qkv.weight"], qkv, M=M, N=3 * Cn, K=Cn, C1=Cn)
            hip.attn_temporal(qkv, 3 * Cn, o, Cn, batch, T, S, heads, scale)
        if legacy:
            hm = self._linear(o, t + ".attn1.to_out.0", M, res1=hm)
            a = hip.layernorm(hm, M, Cn, w[t + ".norm3.weight"], w[t + ".norm3.bias"], addvec=cond.cross[t], addvec_ld=Cn,
                              rows_per_vec=T * S, sum_out=hm, out=a)
        elif cond.P > 1:
            hm = self._linear(o, t + ".attn1.to_out.0", M, res1=hm, stream=tb(4))
            hm = self._cross_attention(t, hm, M, Cn, T * S, heads, cond)
            if not fold_ff:
                a = hip.layernorm(hm, M, Cn, w[t + ".norm3.weight"], w[t + ".norm3.bias"], out=a)
        else:
            hm = self._linear(o, t + ".attn1.to_out.0", M, res1=hm, rowvec=cond.cross[t], rowvec_ld=Cn, rows_per_vec=T * S,
                              stream=tb(4))
            if not fold_ff:
                a = hip.layernorm(hm, M, Cn, w[t + ".norm3.weight"], w[t + ".norm3.bias"], out=a)
        # AlphaBlender: am*hs + (1-am)*(hm + ff(a)); with hs' = hs + emb stored: am*hs = am*hs' - am*emb, and the
        # epilogue's vector enters as alpha * rowvec with alpha = 1 - am  ->  rowvec = -am / (1 - am) * emb
        lnin = hm if fold_ff else None
        # (the blend only feeds proj_out as an MFMA operand: 16-bit in both stream modes)
        if fold_emb:
            hb = self._geglu_ff(a, t + ".ff", M, Cn, ln_input=lnin, alpha=1.0 - am, res1=hm, ldr1=Cn, beta1=1.0 - am, res2=hs,
                                ldr2=Cn, beta2=am, rowvec=cond.pos_emb_blend[p], rowvec_ld=Cn, rows_per_vec=S, stream=False)
        else:
            hb = self._geglu_ff(a, t + ".ff", M, Cn, ln_input=lnin, alpha=1.0 - am, res1=hm, ldr1=Cn, beta1=1.0 - am, res2=hs,
                                ldr2=Cn, beta2=am, stream=False)
        return hb

    # ------------------------------------------------------------------------------------------
    # forward
    # ------------------------------------------------------------------------------------------
    def time_embedding(self, t: float, cond: RequestCond) -> torch.Tensor:
        """silu(emb) rows for every (cfg item, frame): bf16 [Bc*T, E]  (unet:449-487 + resnet.py:343-344)."""
        cfg = self.cfg
        feat = torch.from_numpy(sinusoid(np.full((cond.Bc,), t, np.float32), cfg.block_out_channels[0]))
        temb = self._mlp("time_embedding", feat.to(self.device, self.dtype), cond.Bc)
        out = self._empty(cond.Bc * cfg.num_frames, cfg.time_embed_dim)
        return self.hip.emb_combine(temb, cond.act_emb, cond.noise_emb, cond.Bc, cond.B, cfg.num_frames,
                                    cfg.time_embed_dim, out)

    def forward(self, x_in: torch.Tensor, emb_silu: torch.Tensor, cond: RequestCond, h: int, w_: int) -> torch.Tensor:
        """x_in: bf16 [Bc*T*h*w, 64] (wiw_prep_unet_input); emb_silu: bf16 [Bc*T, E] (time_embedding()).
        Returns fp32 [Bc*T*h*w, 4] (token-major v-prediction)."""
        cfg, hip, w = self.cfg, self.hip, self.w
        T = cfg.num_frames
        ch = cfg.block_out_channels
        n = len(ch)
        L = cfg.layers_per_block
        frames = cond.Bc * T
        M = frames * h * w_
        assert x_in.shape == (M, CIN_PAD) and (h % (1 << (n - 1)) == 0) and (w_ % (1 << (n - 1)) == 0)
        self._ncand = cond.B        # `_conv3` decides from ONE candidate's rows (the bit-exact batch contract)
        temb_all = self._empty(frames, self.temb_total, dtype=torch.float32)
        hip.gemm(emb_silu, w["temb_all.weight"], temb_all, M=frames, N=self.temb_total, K=cfg.time_embed_dim,
                 C1=cfg.time_embed_dim, bias=w["temb_all.bias"], epilogue=EPI_OUT_F32)
        epi_s = EPI_OUT_F32 if self.res32 else 0     # residual-stream outputs (fp32 when residual_fp32)
        x = self._empty(M, ch[0], dtype=self.sdt)
        self._conv3(x_in, "conv_in", x, M=M, N=ch[0], C=CIN_PAD, H=h, W=w_, epilogue=epi_s)
        skips = [(x, ch[0])]
        H, W, C = h, w_, ch[0]
        for i in range(n):
            p = f"down_blocks.{i}"
            has_attn = i < n - 1
            eps = 1e-6 if has_attn else 1e-5
            for j in range(L):
                x = self._res_block(f"{p}.resnets.{j}", x, C, None, 0, ch[i], M, H, W, temb_all, eps)
                C = ch[i]
                if has_attn:
                    x = self._transformer(f"{p}.attentions.{j}", x, C, M, H, W, cfg.num_attention_heads[i], cond)
                skips.append((x, C))
            if i < n - 1:
                H, W, M = H // 2, W // 2, M // 4
                y = self._empty(M, C, dtype=self.sdt)
                q = f"{p}.downsamplers.0.conv"
                # (an fp32 stream enters the convolution as its rounded copy: Downsample2D has no norm in front of it)
                hip.gemm(hip.cast16(x) if self.res32 else x, w[q + ".weight"], y, M=M, N=C, K=9 * C, C1=C, mode=A_CONV3X3_S2,
                         H=H, Wd=W, bias=w[q + ".bias"], epilogue=epi_s | self.kc)
                x = y
                skips.append((x, C))
        x = self._res_block("mid_block.resnets.0", x, C, None, 0, C, M, H, W, temb_all, 1e-5)
        x = self._transformer("mid_block.attentions.0", x, C, M, H, W, cfg.num_attention_heads[-1], cond)
        x = self._res_block("mid_block.resnets.1", x, C, None, 0, C, M, H, W, temb_all, 1e-5)
        rch = list(reversed(ch))
        rheads = list(reversed(cfg.num_attention_heads))
        for i in range(n):
            p = f"up_blocks.{i}"
            for j in range(L + 1):
                sk, Cs = skips.pop()
                x = self._res_block(f"{p}.resnets.{j}", x, C, sk, Cs, rch[i], M, H, W, temb_all, 1e-6)
                C = rch[i]
                if i > 0:
                    x = self._transformer(f"{p}.attentions.{j}", x, C, M, H, W, rheads[i], cond)
            if i < n - 1:
                H, W, M = H * 2, W * 2, M * 4
                y = self._empty(M, C, dtype=self.sdt)
                q = f"{p}.upsamplers.0.conv"
                self._conv3(hip.cast16(x) if self.res32 else x, q, y, M=M, N=C, C=C, H=H, W=W, epilogue=epi_s, mode=A_CONV3X3_UP)
                x = y
        xn = hip.groupnorm(x, C, None, 0, M, H * W, w["conv_norm_out.weight"], w["conv_norm_out.bias"], 1e-5, True)
        out = self._empty(M, cfg.out_channels, dtype=torch.float32)
        hip.gemm(xn, w["conv_out.weight"], out, M=M, N=cfg.out_channels, K=9 * C, C1=C, mode=A_CONV3X3, H=H, Wd=W,
                 bias=w["conv_out.bias"], epilogue=EPI_OUT_F32 | self.kc)
        return out

    # ------------------------------------------------------------------------------------------
    # reference-shaped convenience entry point (same arguments as the reference forward)
    # ------------------------------------------------------------------------------------------
    def __call__(self, sample: torch.Tensor, timestep: float, encoder_hidden_states: torch.Tensor,
                 added_time_ids: torch.Tensor, added_action_ids: torch.Tensor) -> torch.Tensor:
        """sample (Bc,T,8,h,w), encoder_hidden_states (Bc,1,D), added_time_ids (Bc,3), added_action_ids (B,T,Ch)
        -> (Bc,T,4,h,w) fp32.  Layout conversion here is torch plumbing for tests; the served loop
        (pipeline.py) assembles the token-major input with wiw_prep_unet_input instead."""
        Bc, T, Cin, h, w_ = sample.shape
        B = added_action_ids.shape[0]
        assert Bc % B == 0 and T == self.cfg.num_frames
        _require_single_key(encoder_hidden_states, self.max_ctx)
        P = encoder_hidden_states.shape[1]
        ehs = encoder_hidden_states.reshape(Bc, P, -1).to(self.device, torch.float32)
        cond = self.prepare_request(ehs[Bc - B:], added_action_ids.cpu().numpy(),
                                    float(added_time_ids[0, -1]), cfg_batch=(Bc == 2 * B))
        # the general (non-zero uncond embeds) case: overwrite the embeds prepared above
        if Bc == 2 * B and float(ehs[:B].abs().max()) != 0.0:
            raise ValueError("uncond image embeddings must be zeros (pipeline:221-227)")
        x = sample.to(self.device, torch.float32).permute(0, 1, 3, 4, 2).reshape(Bc * T * h * w_, Cin)
        x_in = torch.zeros(Bc * T * h * w_, CIN_PAD, dtype=self.dtype, device=self.device)
        x_in[:, :Cin] = x.to(self.dtype)
        emb = self.time_embedding(float(timestep), cond)
        out = self.forward(x_in, emb, cond, h, w_)
        return out.reshape(Bc, T, h, w_, self.cfg.out_channels).permute(0, 1, 4, 2, 3).contiguous()
