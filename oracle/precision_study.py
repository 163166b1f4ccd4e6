#!/usr/bin/env python
"""Where does a 16-bit evaluation of the served UNet lose its accuracy — and would an fp32 residual stream reach north_star's
1e-3 relative latent error?  ("close the tolerance question instead of arguing it".)

TEST INFRASTRUCTURE (CPU only, imports the oracle).  The fp32 oracle forward (`svd_oracle.unet_forward`, pinned to the
reference) is re-run on the FULL-WIDTH fixture `tests/golden/unet_full_16x32.npz` (weights = seed 4, input and fp32 reference
output produced by the reference itself) with the storage roundings of the HIP path EMULATED: weights rounded to the 16-bit
type, and every tensor the HIP path writes to HBM rounded where it writes it.  Two storage policies:

  all16   every activation tensor in the 16-bit type — what libwiwsvd.so / libwiwsvd_f16.so do today.  Validates the emulation:
          the HIP fp16 build measures 1.21e-3 on this fixture, the bf16 build 9.8e-3 (tests/test_hip_fp16.py,
          tests/test_hip_served_width.py);
  res32   the RESIDUAL STREAM in fp32 — the block outputs x + f(x) of every ResnetBlock / transformer sub-block, the
          AlphaBlender outputs, the skip tensors — while everything consumed as an MFMA operand stays 16-bit (normalised
          activations, q / k / v / attention output, the GEGLU hidden tensor, the convolution outputs inside a ResnetBlock).

Arithmetic inside an operator is fp32 in both (as on the GPU: fp32 accumulation, statistics, softmax).

    python oracle/precision_study.py [fp16|bf16] [all16|res32|weights]...
Prints rms / max relative error of each policy against the reference's fp32 output.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import svd_oracle as O  # noqa: E402
import wiw_amd  # noqa: E402,F401
from wiw_amd.config import UNetConfig  # noqa: E402
from wiw_amd.weights import random_state_dict  # noqa: E402


class Policy:
    """res32: True (the whole residual stream in fp32), False, or a set of categories kept in fp32:
    'tr' = the residual adds inside the transformer layers (proj_in output ... AlphaBlender), 'rb' = ResBlock outputs and
    their AlphaBlender, 'io' = the stream between blocks (conv_in, down / upsampler outputs, transformer output + input)."""

    def __init__(self, dtype, res32):
        self.dtype = dtype
        self.cats = {"tr", "rb", "io"} if res32 is True else (set() if not res32 else set(res32))
        self.res32 = bool(self.cats)
        self.cat = "io"

    def op(self, x):      # a tensor consumed as an MFMA operand / written by an operator inside a block
        return x.to(self.dtype).float() if self.dtype is not None else x

    def res(self, x, cat=None):     # a residual-stream tensor
        keep = (cat or self.cat) in self.cats
        return x if (keep or self.dtype is None) else x.to(self.dtype).float()


def install(pol: Policy):
    """Re-state the block functions of svd_oracle with the storage roundings of the HIP path (same formulas, same order)."""
    op = pol.op
    res = lambda x: pol.res(x, 'io')  # noqa: E731
    res_tr = lambda x: pol.res(x, 'tr')  # noqa: E731
    res_rb = lambda x: pol.res(x, 'rb')  # noqa: E731
    sdget = lambda sd, k: sd[k]  # noqa: E731

    def linear(sd, p, x):
        return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))

    def gn_silu(sd, p, x, eps, silu=True):
        y = F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps)
        return op(F.silu(y) if silu else y)

    def resnet_block_2d(sd, p, x, temb, eps):
        h = gn_silu(sd, p + ".norm1", x, eps)
        h = F.conv2d(h, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
        h = op(h + linear(sd, p + ".time_emb_proj", F.silu(temb))[:, :, None, None])
        h = gn_silu(sd, p + ".norm2", h, eps)
        h = F.conv2d(h, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
        if p + ".conv_shortcut.weight" in sd:     # fused into conv2's implicit GEMM on the HIP path: operands are the 16-bit x
            x = F.conv2d(op(x), sd[p + ".conv_shortcut.weight"], sd[p + ".conv_shortcut.bias"])
        return res_rb(x + h)

    def temporal_resnet_block(sd, p, x, temb, eps):
        h = gn_silu(sd, p + ".norm1", x, eps)
        h = F.conv3d(h, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=(1, 0, 0))
        te = linear(sd, p + ".time_emb_proj", F.silu(temb))
        h = op(h + te.permute(0, 2, 1)[:, :, :, None, None])
        h = gn_silu(sd, p + ".norm2", h, eps)
        h = F.conv3d(h, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=(1, 0, 0))
        return x + h          # blended right away in the conv2 epilogue (fp32), rounded by alpha_blend's caller

    def spatio_temporal_res_block(sd, p, x, temb, T, eps):
        hs = resnet_block_2d(sd, p + ".spatial_res_block", x, temb, eps)
        BF, C, H, W = hs.shape
        B = BF // T
        h5 = hs.reshape(B, T, C, H, W).permute(0, 2, 1, 3, 4)
        ht = temporal_resnet_block(sd, p + ".temporal_res_block", h5, temb.reshape(B, T, -1), eps)
        out = res_rb(O.alpha_blend(sd, p + ".time_mixer", h5, ht))
        return out.permute(0, 2, 1, 3, 4).reshape(BF, C, H, W)

    def layer_norm(sd, p, x):
        return op(F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5))

    def attention(sd, p, x, ctx, heads):
        ctx = x if ctx is None else ctx
        q, k, v = op(F.linear(x, sd[p + ".to_q.weight"])), op(F.linear(ctx, sd[p + ".to_k.weight"])), op(F.linear(ctx, sd[p + ".to_v.weight"]))
        Bn, Sq, C = q.shape
        d = C // heads
        q = q.reshape(Bn, Sq, heads, d).transpose(1, 2)
        k = k.reshape(Bn, -1, heads, d).transpose(1, 2)
        v = v.reshape(Bn, -1, heads, d).transpose(1, 2)
        w = op(torch.softmax((q @ k.transpose(-1, -2)) * (d ** -0.5), dim=-1))       # P is a 16-bit MFMA operand
        o = op((w @ v).transpose(1, 2).reshape(Bn, Sq, C))
        return linear(sd, p + ".to_out.0", o)

    def geglu_ff(sd, p, x):
        h = linear(sd, p + ".net.0.proj", x)
        val, gate = h.chunk(2, dim=-1)
        return linear(sd, p + ".net.2", op(val * F.gelu(gate)))

    def basic_transformer_block(sd, p, x, ehs, heads):
        x = res_tr(x + attention(sd, p + ".attn1", layer_norm(sd, p + ".norm1", x), None, heads))
        x = res_tr(x + attention(sd, p + ".attn2", layer_norm(sd, p + ".norm2", x), ehs, heads))     # (a per-item vector on the GPU)
        x = res_tr(x + geglu_ff(sd, p + ".ff", layer_norm(sd, p + ".norm3", x)))
        return x

    def temporal_transformer_block(sd, p, x, T, time_ctx, heads):
        BF, S, C = x.shape
        B = BF // T
        h = x.reshape(B, T, S, C).permute(0, 2, 1, 3).reshape(B * S, T, C)
        h = res_tr(h + geglu_ff(sd, p + ".ff_in", layer_norm(sd, p + ".norm_in", h)))
        h = res_tr(h + attention(sd, p + ".attn1", layer_norm(sd, p + ".norm1", h), None, heads))
        h = res_tr(h + attention(sd, p + ".attn2", layer_norm(sd, p + ".norm2", h), time_ctx, heads))
        h = h + geglu_ff(sd, p + ".ff", layer_norm(sd, p + ".norm3", h))      # blended in the FF2 epilogue (fp32)
        return h.reshape(B, S, T, C).permute(0, 2, 1, 3).reshape(BF, S, C)

    def transformer_spatio_temporal(sd, p, x, ehs, T, heads):
        BF, C, H, W = x.shape
        B = BF // T
        S = H * W
        tc = ehs.reshape(B, T, -1, ehs.shape[-1])[:, 0]
        tc = tc[:, None].expand(B, S, tc.shape[-2], tc.shape[-1]).reshape(B * S, -1, ehs.shape[-1])
        h = gn_silu(sd, p + ".norm", x, 1e-6, silu=False)
        h = h.permute(0, 2, 3, 1).reshape(BF, S, C)
        h = res_tr(linear(sd, p + ".proj_in", h))
        frame_idx = torch.arange(T).repeat(B)
        emb = O.timestep_mlp(sd, p + ".time_pos_embed", O.timestep_embedding(frame_idx, C))[:, None, :]
        hs = basic_transformer_block(sd, p + ".transformer_blocks.0", h, ehs, heads)
        ht = temporal_transformer_block(sd, p + ".temporal_transformer_blocks.0", res_tr(hs + emb), T, tc, heads)
        h = res_tr(O.alpha_blend(sd, p + ".time_mixer", hs, ht))
        h = linear(sd, p + ".proj_out", op(h) if 'tr' in pol.cats else h)     # an fp32 stream still enters a GEMM as a 16-bit operand
        return res(h.reshape(BF, H, W, C).permute(0, 3, 1, 2) + x)

    saved = {}
    for name, fn in dict(resnet_block_2d=resnet_block_2d, temporal_resnet_block=temporal_resnet_block,
                         spatio_temporal_res_block=spatio_temporal_res_block, basic_transformer_block=basic_transformer_block,
                         temporal_transformer_block=temporal_transformer_block,
                         transformer_spatio_temporal=transformer_spatio_temporal).items():
        saved[name] = getattr(O, name)
        setattr(O, name, fn)
    # convolutions that sit directly in unet_forward (conv_in, down / upsamplers, conv_out) read and write the stream
    conv2d = F.conv2d

    def conv2d_stream(x, w, b=None, **kw):
        return res(conv2d(op(x) if pol.cats else x, w, b, **kw))
    saved["_conv2d"] = conv2d
    O.F.conv2d = conv2d_stream
    _ = sdget
    return saved


def uninstall(saved):
    O.F.conv2d = saved.pop("_conv2d")
    for k, v in saved.items():
        setattr(O, k, v)


def main():
    args = sys.argv[1:]
    dtype = torch.bfloat16 if "bf16" in args else torch.float16
    which = [a for a in args if a in ("all16", "res32", "weights", "tr", "rb", "io", "tr+rb", "tr+io", "rb+io")] or ["weights", "all16", "res32"]
    g = np.load(os.path.join(ROOT, "tests", "golden", "unet_full_16x32.npz"), allow_pickle=True)
    cfg = UNetConfig()
    sd = {k: torch.from_numpy(v).to(dtype).float() for k, v in random_state_dict(cfg, int(g["weight_seed"])).items()}
    ref = torch.from_numpy(g["out"])
    torch.set_num_threads(8)
    ins = (torch.from_numpy(g["sample"]), torch.tensor(float(g["timestep"])), torch.from_numpy(g["ehs"]),
           torch.from_numpy(g["added_time_ids"]), torch.from_numpy(g["action_ids"]))

    def report(name, out):
        err = (out - ref)
        rms = float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
        mx = float(err.abs().max() / ref.abs().max())
        print(f"{str(dtype).split('.')[-1]:9s} {name:8s} rms_rel = {rms:.3e}   max_rel = {mx:.3e}", flush=True)
        return rms

    with torch.no_grad():
        for w in which:
            if w == "weights":          # fp32 arithmetic and storage on the ROUNDED weights: the floor of any policy
                report("weights", O.unet_forward(sd, cfg.as_dict(), *ins))
                continue
            saved = install(Policy(dtype, res32=True if w == "res32" else (False if w == "all16" else w.split("+"))))
            try:
                report(w, O.unet_forward(sd, cfg.as_dict(), *ins))
            finally:
                uninstall(saved)


if __name__ == "__main__":
    main()
