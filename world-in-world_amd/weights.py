"""Weight schema of the served UNet and deterministic random initialisation.

The on-disk schema is the diffusers safetensors layout of
`unet/diffusion_pytorch_model.safetensors` (reference module tree:
FTsvd/diffusers-private/diffusers/models/unets/unet_spatio_temporal_condition.py:130-283 and
unet_3d_blocks.py:1043-1674).  `state_dict_spec` enumerates it from a `UNetConfig` so that
(1) a checkpoint can be validated before it is re-laid-out for the kernels and
(2) parity fixtures / the benchmark can draw seeded random weights of the exact architecture
(no checkpoint is available offline).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np

from .config import UNetConfig

Spec = "OrderedDict[str, Tuple[int, ...]]"


def _lin(spec, p, cin, cout, bias=True):
    spec[p + ".weight"] = (cout, cin)
    if bias:
        spec[p + ".bias"] = (cout,)


def _norm(spec, p, c):
    spec[p + ".weight"] = (c,)
    spec[p + ".bias"] = (c,)


def _tmlp(spec, p, cin, hidden, cout):
    _lin(spec, p + ".linear_1", cin, hidden)
    _lin(spec, p + ".linear_2", hidden, cout)


def _res(spec, p, cin, cout, temb):
    s = p + ".spatial_res_block"
    _norm(spec, s + ".norm1", cin)
    spec[s + ".conv1.weight"] = (cout, cin, 3, 3)
    spec[s + ".conv1.bias"] = (cout,)
    _lin(spec, s + ".time_emb_proj", temb, cout)
    _norm(spec, s + ".norm2", cout)
    spec[s + ".conv2.weight"] = (cout, cout, 3, 3)
    spec[s + ".conv2.bias"] = (cout,)
    if cin != cout:
        spec[s + ".conv_shortcut.weight"] = (cout, cin, 1, 1)
        spec[s + ".conv_shortcut.bias"] = (cout,)
    t = p + ".temporal_res_block"
    _norm(spec, t + ".norm1", cout)
    spec[t + ".conv1.weight"] = (cout, cout, 3, 1, 1)
    spec[t + ".conv1.bias"] = (cout,)
    _lin(spec, t + ".time_emb_proj", temb, cout)
    _norm(spec, t + ".norm2", cout)
    spec[t + ".conv2.weight"] = (cout, cout, 3, 1, 1)
    spec[t + ".conv2.bias"] = (cout,)
    spec[p + ".time_mixer.mix_factor"] = (1,)


def _attn(spec, p, c, ctx=None):
    _lin(spec, p + ".to_q", c, c, bias=False)
    _lin(spec, p + ".to_k", ctx or c, c, bias=False)
    _lin(spec, p + ".to_v", ctx or c, c, bias=False)
    _lin(spec, p + ".to_out.0", c, c)


def _ff(spec, p, c):
    _lin(spec, p + ".net.0.proj", c, 8 * c)
    _lin(spec, p + ".net.2", 4 * c, c)


def _transformer(spec, p, c, ctx):
    _norm(spec, p + ".norm", c)
    _lin(spec, p + ".proj_in", c, c)
    b = p + ".transformer_blocks.0"
    _norm(spec, b + ".norm1", c)
    _attn(spec, b + ".attn1", c)
    _norm(spec, b + ".norm2", c)
    _attn(spec, b + ".attn2", c, ctx)
    _norm(spec, b + ".norm3", c)
    _ff(spec, b + ".ff", c)
    t = p + ".temporal_transformer_blocks.0"
    _norm(spec, t + ".norm_in", c)
    _ff(spec, t + ".ff_in", c)
    _norm(spec, t + ".norm1", c)
    _attn(spec, t + ".attn1", c)
    _norm(spec, t + ".norm2", c)
    _attn(spec, t + ".attn2", c, ctx)
    _norm(spec, t + ".norm3", c)
    _ff(spec, t + ".ff", c)
    _tmlp(spec, p + ".time_pos_embed", c, 4 * c, c)
    spec[p + ".time_mixer.mix_factor"] = (1,)
    _lin(spec, p + ".proj_out", c, c)


def state_dict_spec(cfg: UNetConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """name -> shape for every tensor of the reference module's `state_dict()`."""
    spec: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    ch = cfg.block_out_channels
    temb = cfg.time_embed_dim
    ctx = cfg.cross_attention_dim
    L = cfg.layers_per_block
    n = len(ch)
    spec["conv_in.weight"] = (ch[0], cfg.in_channels, 3, 3)
    spec["conv_in.bias"] = (ch[0],)
    _tmlp(spec, "time_embedding", ch[0], temb, temb)
    _tmlp(spec, "add_embedding", cfg.projection_class_embeddings_input_dim, temb, temb)  # dead for micro_cond
    out_c = ch[0]
    for i in range(n):
        in_c, out_c = out_c, ch[i]
        p = f"down_blocks.{i}"
        for j in range(L):
            _res(spec, f"{p}.resnets.{j}", in_c if j == 0 else out_c, out_c, temb)
            if i < n - 1:
                _transformer(spec, f"{p}.attentions.{j}", out_c, ctx)
        if i < n - 1:
            spec[f"{p}.downsamplers.0.conv.weight"] = (out_c, out_c, 3, 3)
            spec[f"{p}.downsamplers.0.conv.bias"] = (out_c,)
    _res(spec, "mid_block.resnets.0", ch[-1], ch[-1], temb)
    _transformer(spec, "mid_block.attentions.0", ch[-1], ctx)
    _res(spec, "mid_block.resnets.1", ch[-1], ch[-1], temb)
    rch = list(reversed(ch))
    out_c = rch[0]
    for i in range(n):
        prev_out, out_c = out_c, rch[i]
        in_c = rch[min(i + 1, n - 1)]
        p = f"up_blocks.{i}"
        for j in range(L + 1):
            skip_c = in_c if j == L else out_c
            res_in = prev_out if j == 0 else out_c
            _res(spec, f"{p}.resnets.{j}", res_in + skip_c, out_c, temb)
            if i > 0:
                _transformer(spec, f"{p}.attentions.{j}", out_c, ctx)
        if i < n - 1:
            spec[f"{p}.upsamplers.0.conv.weight"] = (out_c, out_c, 3, 3)
            spec[f"{p}.upsamplers.0.conv.bias"] = (out_c,)
    _norm(spec, "conv_norm_out", ch[0])
    spec["conv_out.weight"] = (cfg.out_channels, ch[0], 3, 3)
    spec["conv_out.bias"] = (cfg.out_channels,)
    _lin(spec, "add_action_proj.proj", cfg.action_input_channel * 12, 256)
    _tmlp(spec, "add_embedding_action", 256, temb, temb)
    _tmlp(spec, "add_embedding_noise", 256, temb, temb)
    return spec


def skip_channels(cfg: UNetConfig):
    """Channel counts of the 12 skip tensors in push order (unet:514-531)."""
    ch = cfg.block_out_channels
    out = [ch[0]]
    for i, c in enumerate(ch):
        out += [c] * cfg.layers_per_block
        if i < len(ch) - 1:
            out.append(c)
    return out


def _fan_in(shape) -> int:
    f = 1
    for s in shape[1:]:
        f *= s
    return max(f, 1)


def random_state_dict(cfg: UNetConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    """Deterministic (numpy RandomState, machine-independent) fp32 weights for parity fixtures.

    Weights/biases ~ U(-1/sqrt(fan_in), +1/sqrt(fan_in)) (torch's default Linear/Conv bound),
    norm weight = 1 + 0.1 N(0,1), norm bias = 0.1 N(0,1), mix_factor ~ U(-1,1)."""
    rs = np.random.RandomState(seed)
    sd: Dict[str, np.ndarray] = {}
    spec = state_dict_spec(cfg)
    for name, shape in spec.items():
        if name.endswith("mix_factor"):
            v = rs.uniform(-1.0, 1.0, size=shape)
        elif len(shape) == 1 and (".norm" in name or name.startswith("conv_norm_out")) and name.endswith(".weight"):
            v = 1.0 + 0.1 * rs.standard_normal(shape)
        elif len(shape) == 1 and (".norm" in name or name.startswith("conv_norm_out")):
            v = 0.1 * rs.standard_normal(shape)
        else:
            wname = name[:-5] + ".weight" if name.endswith(".bias") else name
            bound = 1.0 / np.sqrt(_fan_in(spec.get(wname, shape)))
            v = rs.uniform(-bound, bound, size=shape)
        sd[name] = v.astype(np.float32)
    return sd


def random_state_dict_torch(cfg: UNetConfig, seed: int, device, dtype):
    """Fast on-device random init for the benchmark (no cross-machine determinism needed)."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    sd = {}
    spec = state_dict_spec(cfg)
    for name, shape in spec.items():
        if name.endswith("mix_factor"):
            v = torch.rand(shape, generator=g, device=device) * 2 - 1
        elif len(shape) == 1 and (".norm" in name or name.startswith("conv_norm_out")):
            v = torch.randn(shape, generator=g, device=device) * 0.1
            if name.endswith(".weight"):
                v = v + 1.0
        else:
            wname = name[:-5] + ".weight" if name.endswith(".bias") else name
            bound = 1.0 / float(np.sqrt(_fan_in(spec.get(wname, shape))))
            v = (torch.rand(shape, generator=g, device=device) * 2 - 1) * bound
        sd[name] = v.to(dtype)
    return sd


def load_safetensors(path: str) -> Dict[str, "np.ndarray"]:
    """Load a diffusers `diffusion_pytorch_model*.safetensors` file into torch tensors."""
    from safetensors.torch import load_file

    return load_file(path)


def validate_state_dict(cfg: UNetConfig, sd) -> None:
    """Raise if `sd` does not carry exactly the served architecture (missing / mis-shaped tensors).
    Extra `add_embedding.*` tensors are tolerated (dead weights for micro_cond)."""
    spec = state_dict_spec(cfg)
    missing = [k for k in spec if k not in sd and not k.startswith("add_embedding.")]
    if missing:
        raise KeyError(f"checkpoint is missing {len(missing)} tensors, e.g. {missing[:5]}")
    for k, shp in spec.items():
        if k in sd and tuple(sd[k].shape) != tuple(shp):
            raise ValueError(f"{k}: expected shape {shp}, got {tuple(sd[k].shape)}")
