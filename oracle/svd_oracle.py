"""CPU oracle for the SVD denoising hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

A plain torch-fp32 (CPU) functional restatement of the reference's algorithm for the path
`StableVideoDiffusionPipeline.__call__` -> `UNetSpatioTemporalConditionModel.forward` ->
`EulerDiscreteScheduler.step` (World-In-World, vendored diffusers 0.31 fork, `micro_cond`
action strategy).  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
leg may import this module; the product path (`world-in-world_amd/`) never does.

Pinning: the reference holds NO golden vectors or tests for this path, so this
restatement is pinned against outputs of the reference itself, imported in the build container
by `oracle/make_golden.py` (fixtures committed under `tests/golden/`, checked by
`tests/test_oracle_golden.py`).  The arithmetic below the reference's Python (ATen conv/addmm/
SDPA/group_norm) is third-party: parity there is "unpinned by the reference" and pinned by the
fixtures generated with torch 2.10 (versions are recorded in each fixture).

Citations use the prefixes of SURVEY.md: `dp/` = FTsvd/diffusers-private/diffusers/,
unet: = dp/models/unets/unet_spatio_temporal_condition.py,
pipeline: = dp/pipelines/stable_video_diffusion/pipeline_stable_video_diffusion.py,
scheduler: = dp/schedulers/scheduling_euler_discrete.py.

All tensors here are in the REFERENCE layouts (NCHW / (B,T,C,H,W)); weights are the diffusers
state-dict (`name -> tensor`, Linear weights [out,in], conv weights OIHW).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

# ----------------------------------------------------------------------------------------------
# configuration (class defaults of unet:72-97 + the fork's runtime kwargs, eval_inference.py:116-125)
# ----------------------------------------------------------------------------------------------
DEFAULT_UNET_CONFIG = dict(
    in_channels=8,
    out_channels=4,
    block_out_channels=(320, 640, 1280, 1280),
    num_attention_heads=(5, 10, 20, 20),
    layers_per_block=2,
    cross_attention_dim=1024,
    addition_time_embed_dim=256,
    num_frames=14,
    action_input_channel=14,
)


def _cfg(cfg: Optional[dict]) -> dict:
    c = dict(DEFAULT_UNET_CONFIG)
    if cfg:
        c.update(cfg)
    return c


# ----------------------------------------------------------------------------------------------
# embeddings
# ----------------------------------------------------------------------------------------------
def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """dp/models/embeddings.py:27-78 with flip_sin_to_cos=True, downscale_freq_shift=0, scale=1
    (how every `Timesteps` on this path is constructed: unet:141,146; transformer_temporal.py:269)."""
    assert t.dim() == 1
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(half, dtype=torch.float32) / half
    emb = t[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)  # flipped: [cos | sin]


def linear(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def timestep_mlp(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """`TimestepEmbedding.forward` dp/models/embeddings.py:804-816 (linear_1 -> SiLU -> linear_2)."""
    return linear(sd, p + ".linear_2", F.silu(linear(sd, p + ".linear_1", x)))


def action_embedder(sd: SD, x: torch.Tensor) -> torch.Tensor:
    """`ActionEmbedder_.forward` dp/models/embeddings.py:914-944: 12 Fourier features per entry,
    stacked on the LAST dim then flattened channel-major, then Linear(input_channel*12 -> 256)."""
    bs, n, m = x.shape
    feats = []
    for k in (1.0, 2.0, 4.0, 6.0, 8.0, 10.0):
        feats += [torch.cos(k * x), torch.sin(k * x)]
    f = torch.stack(feats, dim=-1).reshape(bs, n, m * 12)
    return linear(sd, "add_action_proj.proj", f)


def conditioning_embedding(sd: SD, cfg: dict, t: torch.Tensor, added_time_ids: torch.Tensor,
                           action_ids: torch.Tensor, batch: int, reference_quirk: bool = False) -> torch.Tensor:
    """emb of unet:449-487 for `micro_cond` under torch.no_grad (inference).

    batch = CFG batch (2*B: [uncond_0..uncond_{B-1}, cond_0..cond_{B-1}], pipeline:589).
    Returns (batch*T, 4*C0).  `aug_emb` (unet:458-461) is dead for micro_cond (overwritten :482).

    reference_quirk=False (the contract): sample row (i, t) gets the action row of
    candidate i % B.  reference_quirk=True reproduces unet:479-481 literally
    (`repeat(1,2,1)` then `'b t c -> (b t) c'`), which equals the contract only for B == 1.
    """
    c0 = cfg["block_out_channels"][0]
    T = cfg["num_frames"]
    t_emb = timestep_embedding(t.reshape(1).expand(batch), c0)  # unet:447-449
    emb_time = timestep_mlp(sd, "time_embedding", t_emb)  # (batch, 4*c0)
    emb_time = emb_time.repeat_interleave(T, dim=0)  # unet:465
    act = timestep_mlp(sd, "add_embedding_action", action_embedder(sd, action_ids.float()))  # (B,T,E) unet:472, 274-280
    B = act.shape[0]
    if reference_quirk:
        act_rows = act.repeat(1, 2, 1).reshape(B * 2 * T, -1)  # unet:479-481
    else:
        assert batch % B == 0
        act_rows = act.repeat(batch // B, 1, 1).reshape(batch * T, -1)
    noise = timestep_embedding(added_time_ids[:, -1].flatten(), cfg["addition_time_embed_dim"])  # unet:484
    noise = timestep_mlp(sd, "add_embedding_noise", noise).repeat_interleave(T, dim=0)  # unet:485-486
    return emb_time + act_rows + noise  # unet:482, 487


# ----------------------------------------------------------------------------------------------
# resnet blocks
# ----------------------------------------------------------------------------------------------
def group_norm(sd: SD, p: str, x: torch.Tensor, eps: float) -> torch.Tensor:
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps)


def resnet_block_2d(sd: SD, p: str, x: torch.Tensor, temb: torch.Tensor, eps: float) -> torch.Tensor:
    """`ResnetBlock2D.forward` dp/models/resnet.py:320-373 (time_embedding_norm='default',
    output_scale_factor=1, no up/down)."""
    h = F.silu(group_norm(sd, p + ".norm1", x, eps))
    h = F.conv2d(h, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    h = h + linear(sd, p + ".time_emb_proj", F.silu(temb))[:, :, None, None]
    h = F.silu(group_norm(sd, p + ".norm2", h, eps))
    h = F.conv2d(h, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    if p + ".conv_shortcut.weight" in sd:
        x = F.conv2d(x, sd[p + ".conv_shortcut.weight"], sd[p + ".conv_shortcut.bias"])
    return x + h


def temporal_resnet_block(sd: SD, p: str, x: torch.Tensor, temb: torch.Tensor, eps: float) -> torch.Tensor:
    """`TemporalResnetBlock.forward` dp/models/resnet.py:608-631.  x: (B,C,T,H,W), temb: (B,T,E)."""
    h = F.silu(group_norm(sd, p + ".norm1", x, eps))
    h = F.conv3d(h, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=(1, 0, 0))
    te = linear(sd, p + ".time_emb_proj", F.silu(temb))  # (B,T,C)
    h = h + te.permute(0, 2, 1)[:, :, :, None, None]
    h = F.silu(group_norm(sd, p + ".norm2", h, eps))
    h = F.conv3d(h, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=(1, 0, 0))
    return x + h


def alpha_blend(sd: SD, p: str, x_spatial: torch.Tensor, x_temporal: torch.Tensor) -> torch.Tensor:
    """`AlphaBlender` dp/models/resnet.py:784-797 with image_only_indicator == 0 (unet:512):
    alpha = sigmoid(mix_factor), scalar."""
    a = torch.sigmoid(sd[p + ".mix_factor"]).reshape(())
    return a * x_spatial + (1.0 - a) * x_temporal


def spatio_temporal_res_block(sd: SD, p: str, x: torch.Tensor, temb: torch.Tensor, T: int, eps: float) -> torch.Tensor:
    """`SpatioTemporalResBlock.forward` dp/models/resnet.py:686-716."""
    hs = resnet_block_2d(sd, p + ".spatial_res_block", x, temb, eps)
    BF, C, H, W = hs.shape
    B = BF // T
    h5 = hs.reshape(B, T, C, H, W).permute(0, 2, 1, 3, 4)
    ht = temporal_resnet_block(sd, p + ".temporal_res_block", h5, temb.reshape(B, T, -1), eps)
    out = alpha_blend(sd, p + ".time_mixer", h5, ht)
    return out.permute(0, 2, 1, 3, 4).reshape(BF, C, H, W)


# ----------------------------------------------------------------------------------------------
# transformer blocks
# ----------------------------------------------------------------------------------------------
def layer_norm(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def attention(sd: SD, p: str, x: torch.Tensor, ctx: Optional[torch.Tensor], heads: int) -> torch.Tensor:
    """`Attention` + `AttnProcessor2_0.__call__` dp/models/attention_processor.py:2321-2403:
    to_q/k/v without bias, scale 1/sqrt(dim_head), no mask, to_out.0 with bias."""
    ctx = x if ctx is None else ctx
    q = F.linear(x, sd[p + ".to_q.weight"])
    k = F.linear(ctx, sd[p + ".to_k.weight"])
    v = F.linear(ctx, sd[p + ".to_v.weight"])
    Bn, Sq, C = q.shape
    d = C // heads
    q = q.reshape(Bn, Sq, heads, d).transpose(1, 2)
    k = k.reshape(Bn, -1, heads, d).transpose(1, 2)
    v = v.reshape(Bn, -1, heads, d).transpose(1, 2)
    w = torch.softmax((q @ k.transpose(-1, -2)) * (d ** -0.5), dim=-1)
    o = (w @ v).transpose(1, 2).reshape(Bn, Sq, C)
    return linear(sd, p + ".to_out.0", o)


def geglu_ff(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """`FeedForward` dp/models/attention.py:1185-1243 with `GEGLU` dp/models/activations.py:117-123
    (value half first, gate half second, exact erf GELU)."""
    h = linear(sd, p + ".net.0.proj", x)
    val, gate = h.chunk(2, dim=-1)
    return linear(sd, p + ".net.2", val * F.gelu(gate))


def basic_transformer_block(sd: SD, p: str, x: torch.Tensor, ehs: torch.Tensor, heads: int) -> torch.Tensor:
    """`BasicTransformerBlock.forward` dp/models/attention.py:462-582, norm_type='layer_norm'."""
    x = x + attention(sd, p + ".attn1", layer_norm(sd, p + ".norm1", x), None, heads)
    x = x + attention(sd, p + ".attn2", layer_norm(sd, p + ".norm2", x), ehs, heads)
    x = x + geglu_ff(sd, p + ".ff", layer_norm(sd, p + ".norm3", x))
    return x


def temporal_transformer_block(sd: SD, p: str, x: torch.Tensor, T: int, time_ctx: torch.Tensor, heads: int) -> torch.Tensor:
    """`TemporalBasicTransformerBlock.forward` dp/models/attention.py:707-762 (is_res=True)."""
    BF, S, C = x.shape
    B = BF // T
    h = x.reshape(B, T, S, C).permute(0, 2, 1, 3).reshape(B * S, T, C)
    h = h + geglu_ff(sd, p + ".ff_in", layer_norm(sd, p + ".norm_in", h))
    h = h + attention(sd, p + ".attn1", layer_norm(sd, p + ".norm1", h), None, heads)
    h = h + attention(sd, p + ".attn2", layer_norm(sd, p + ".norm2", h), time_ctx, heads)
    h = h + geglu_ff(sd, p + ".ff", layer_norm(sd, p + ".norm3", h))
    return h.reshape(B, S, T, C).permute(0, 2, 1, 3).reshape(BF, S, C)


def transformer_spatio_temporal(sd: SD, p: str, x: torch.Tensor, ehs: torch.Tensor, T: int, heads: int) -> torch.Tensor:
    """`TransformerSpatioTemporalModel.forward` dp/models/transformers/transformer_temporal.py:279-382
    (one layer, unet transformer_layers_per_block=1).  x: (BF,C,H,W); ehs: (BF,1,Dctx)."""
    BF, C, H, W = x.shape
    B = BF // T
    S = H * W
    # time_context = first frame's tokens, broadcast over every spatial site (:313-320)
    tc = ehs.reshape(B, T, -1, ehs.shape[-1])[:, 0]
    tc = tc[:, None].expand(B, S, tc.shape[-2], tc.shape[-1]).reshape(B * S, -1, ehs.shape[-1])
    h = group_norm(sd, p + ".norm", x, 1e-6)
    h = h.permute(0, 2, 3, 1).reshape(BF, S, C)
    h = linear(sd, p + ".proj_in", h)
    frame_idx = torch.arange(T).repeat(B)
    emb = timestep_mlp(sd, p + ".time_pos_embed", timestep_embedding(frame_idx, C))[:, None, :]
    hs = basic_transformer_block(sd, p + ".transformer_blocks.0", h, ehs, heads)
    ht = temporal_transformer_block(sd, p + ".temporal_transformer_blocks.0", hs + emb, T, tc, heads)
    h = alpha_blend(sd, p + ".time_mixer", hs, ht)
    h = linear(sd, p + ".proj_out", h)
    return h.reshape(BF, H, W, C).permute(0, 3, 1, 2) + x


# ----------------------------------------------------------------------------------------------
# UNet
# ----------------------------------------------------------------------------------------------
def unet_forward(sd: SD, cfg: Optional[dict], sample: torch.Tensor, t: torch.Tensor, ehs: torch.Tensor,
                 added_time_ids: torch.Tensor, action_ids: torch.Tensor, reference_quirk: bool = False) -> torch.Tensor:
    """`UNetSpatioTemporalConditionModel.forward` unet:402-575, micro_cond.

    sample (Bc,T,8,h,w); ehs (Bc,1,Dctx); added_time_ids (Bc,3); action_ids (B,T,T); Bc = 2B under CFG.
    Returns (Bc,T,4,h,w).
    """
    cfg = _cfg(cfg)
    sd = {k: v.float() for k, v in sd.items()}
    Bc, T = sample.shape[:2]
    chans = cfg["block_out_channels"]
    heads = cfg["num_attention_heads"]
    L = cfg["layers_per_block"]
    emb = conditioning_embedding(sd, cfg, torch.as_tensor(t, dtype=torch.float32), added_time_ids.float(),
                                 action_ids, Bc, reference_quirk)
    x = sample.float().flatten(0, 1)
    ehs_f = ehs.float().repeat_interleave(T, dim=0)  # unet:507
    x = F.conv2d(x, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    skips: List[torch.Tensor] = [x]
    n = len(chans)
    for i in range(n):
        p = f"down_blocks.{i}"
        has_attn = i < n - 1  # CrossAttnDown x3 then DownBlockSpatioTemporal (unet:76-81)
        eps = 1e-6 if has_attn else 1e-5  # unet_3d_blocks.py:1308 vs :1202
        for j in range(L):
            x = spatio_temporal_res_block(sd, f"{p}.resnets.{j}", x, emb, T, eps)
            if has_attn:
                x = transformer_spatio_temporal(sd, f"{p}.attentions.{j}", x, ehs_f, T, heads[i])
            skips.append(x)
        if i < n - 1:  # Downsample2D: conv3x3 stride 2 pad 1 (dp/models/downsampling.py:132-150)
            x = F.conv2d(x, sd[f"{p}.downsamplers.0.conv.weight"], sd[f"{p}.downsamplers.0.conv.bias"], stride=2, padding=1)
            skips.append(x)
    x = spatio_temporal_res_block(sd, "mid_block.resnets.0", x, emb, T, 1e-5)
    x = transformer_spatio_temporal(sd, "mid_block.attentions.0", x, ehs_f, T, heads[-1])
    x = spatio_temporal_res_block(sd, "mid_block.resnets.1", x, emb, T, 1e-5)
    rheads = list(reversed(heads))
    for i in range(n):
        p = f"up_blocks.{i}"
        has_attn = i > 0
        for j in range(L + 1):
            x = torch.cat([x, skips.pop()], dim=1)  # unet_3d_blocks.py:1482, 1612
            x = spatio_temporal_res_block(sd, f"{p}.resnets.{j}", x, emb, T, 1e-6)  # get_up_block drops resnet_eps (§9.4)
            if has_attn:
                x = transformer_spatio_temporal(sd, f"{p}.attentions.{j}", x, ehs_f, T, rheads[i])
        if i < n - 1:  # Upsample2D: nearest x2 then conv3x3 (dp/models/upsampling.py:142-186)
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = F.conv2d(x, sd[f"{p}.upsamplers.0.conv.weight"], sd[f"{p}.upsamplers.0.conv.bias"], padding=1)
    x = F.silu(group_norm(sd, "conv_norm_out", x, 1e-5))
    x = F.conv2d(x, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)
    return x.reshape(Bc, T, *x.shape[1:])


# ----------------------------------------------------------------------------------------------
# scheduler (EulerDiscreteScheduler with the SVD config)
# ----------------------------------------------------------------------------------------------
def karras_sigmas(num_steps: int, sigma_min: float = 0.002, sigma_max: float = 700.0, rho: float = 7.0) -> np.ndarray:
    """scheduler:476-499 (`_convert_to_karras`) + :438-450: float64 ramp -> float32, append 0."""
    ramp = np.linspace(0, 1, num_steps)
    min_inv = sigma_min ** (1 / rho)
    max_inv = sigma_max ** (1 / rho)
    sig = (max_inv + ramp * (min_inv - max_inv)) ** rho
    return np.concatenate([sig.astype(np.float32), np.zeros(1, dtype=np.float32)])


def sigma_to_timestep(sigmas: np.ndarray) -> np.ndarray:
    """timestep_type='continuous': t = 0.25*ln(sigma) (scheduler:443-444), float32."""
    return np.array([0.25 * math.log(float(s)) for s in sigmas[:-1]], dtype=np.float32)


def init_noise_sigma(sigmas: np.ndarray) -> float:
    """scheduler:262-268 for timestep_spacing='leading': sqrt(sigma_max^2 + 1)."""
    return float((float(sigmas.max()) ** 2 + 1) ** 0.5)


def euler_step(v: torch.Tensor, x: torch.Tensor, sigma: float, sigma_next: float) -> torch.Tensor:
    """scheduler:635-673, prediction_type='v_prediction', gamma=0; fp32 math."""
    x = x.float()
    x0 = v.float() * (-sigma / (sigma ** 2 + 1) ** 0.5) + x / (sigma ** 2 + 1)
    return x + (x - x0) / sigma * (sigma_next - sigma)


# ----------------------------------------------------------------------------------------------
# action ids & noise rotation
# ----------------------------------------------------------------------------------------------
def action_ids_idx_encode(actions: np.ndarray) -> np.ndarray:
    """utils/svd_utils.py:594-632: frame i = [stop(4), a1..ai, 0...]; first action forced to 4."""
    actions = np.asarray(actions)
    B, L = actions.shape
    out = np.zeros((B, L, L), dtype=np.float32)
    for b in range(B):
        frame = np.zeros(L, dtype=np.float32)
        for i in range(L):
            frame[i] = 4 if i == 0 else actions[b, i]
            out[b, i] = frame
    return out


# ---- manipulation actions (utils/svd_utils.py:357-567), restated row by row as the reference loops over them
_SCENE_BOUNDS = np.array([-0.3, -0.5, 0.6, 0.7, 0.5, 1.6])     # utils/svd_utils.py:15


def quaternion_to_rotmatrix(q) -> np.ndarray:
    """scipy Rotation.from_quat((x, y, z, w)).as_matrix() (utils/svd_utils.py:357-375): normalise, then the textbook matrix."""
    x, y, z, w = (np.asarray(q, dtype=np.float64) / np.linalg.norm(np.asarray(q, dtype=np.float64)))
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def normalize_action(rel_xyz, rel_r6, gripper, low=-2 * np.pi, high=2 * np.pi):
    """utils/svd_utils.py:499-539."""
    span = _SCENE_BOUNDS[3:] - _SCENE_BOUNDS[:3]
    xyz01 = (np.clip(np.asarray(rel_xyz) / np.maximum(span, 1e-8), -1.0, 1.0) + 1.0) * 0.5
    r601 = (np.clip(rel_r6, -1.0, 1.0) + 1.0) * 0.5
    g01 = np.clip(gripper, 0.0, 1.0)
    return xyz01 * (high - low) + low, r601 * (high / 2 - low / 2) + low / 2, g01 * (high / 2 - low / 2) + low / 2


def manip_action_ids(actions: np.ndarray, absolute: bool = True) -> np.ndarray:
    """get_action_ids for 3-D actions (utils/svd_utils.py:544-567): get_norm_actions (:418-457) / get_relative_norm_actions
    (:459-497) per batch item; (B, T, 8) -> (B, T, 10) float32."""
    acts = np.asarray(actions, dtype=np.float32).astype(np.float64)
    out = []
    center = 0.5 * (_SCENE_BOUNDS[:3] + _SCENE_BOUNDS[3:])
    for a in acts:
        xyz, grip = a[:, :3], a[:, 7]
        R = np.stack([quaternion_to_rotmatrix(q) for q in a[:, 3:7]])
        rows = np.zeros((a.shape[0], 10))
        for i in range(a.shape[0]):
            if absolute:
                nx, nr, ng = normalize_action(2.0 * (xyz[i] - center), R[i][:, :2].reshape(6), grip[i])
            elif i == 0:
                continue
            else:
                rel = R[i - 1].T @ (xyz[i] - xyz[i - 1])
                nx, nr, ng = normalize_action(rel, (R[i - 1].T @ R[i])[:, :2].reshape(6), grip[i])
            rows[i] = np.concatenate([nx, nr, [ng]])
        out.append(rows)
    return np.stack(out).astype(np.float32)


def action_encode_positional(ids: np.ndarray) -> np.ndarray:
    """utils/svd_utils.py:570-592: (B, T, L) -> (B, T, T + L - 1)."""
    B, T, L = ids.shape
    out = np.zeros((B, T, T + L - 1), dtype=np.float32)
    for b in range(B):
        for i in range(T):
            out[b, i, i:i + L] = ids[b, i]
    return out


def decode_action_seq_frames(action_ids: np.ndarray) -> np.ndarray:
    """pipeline:826-844: the diagonal."""
    return np.diagonal(np.asarray(action_ids), axis1=-2, axis2=-1)


def rotate_latent_noise(noise: torch.Tensor, actions: np.ndarray) -> torch.Tensor:
    """pipeline:750-786 (`sample_latent_noise` after the randn draw) + :795-823: for action 2 (turn
    left) frame i := roll(frame i-1, +W/16); action 3 (turn right): roll(frame i-1, -W/16)."""
    noise = noise.clone()
    B, Fr = actions.shape
    W = noise.shape[-1]
    assert W % 16 == 0
    for b in range(B):
        for i in range(1, Fr):
            a = actions[b, i]
            if a == 2:
                noise[b, i] = torch.roll(noise[b, i - 1].clone(), W // 16, dims=-1)
            elif a == 3:
                noise[b, i] = torch.roll(noise[b, i - 1].clone(), -(W // 16), dims=-1)
    return noise


# ----------------------------------------------------------------------------------------------
# denoising loop (pipeline:555-622)
# ----------------------------------------------------------------------------------------------
def denoise(sd: SD, cfg: Optional[dict], image_latents: torch.Tensor, image_embeddings: torch.Tensor,
            noise: torch.Tensor, actions: np.ndarray, num_steps: int = 25, fps: int = 7, motion_bucket_id: int = 127,
            noise_aug_strength: float = 0.02, min_guidance: float = 1.0, max_guidance: float = 3.0,
            store_model_dtype: Optional[torch.dtype] = None, return_trajectory: bool = False):
    """The hot loop of pipeline:555-622 for B candidates, each evaluated as an independent B=1 run
    (the contract of SURVEY.md §9.2).

    image_latents (B,4,h,w): VAE mode() of the noisy cond image (cond half; uncond half is zeros,
    pipeline:244-250).  image_embeddings (B,1,Dctx): CLIP embeds (uncond half zeros, :221-227).
    noise (B,T,4,h,w): unit Gaussian draw BEFORE rotation and init_noise_sigma scaling.
    actions (B,T) int, or (B,T,8) float for manipulation.  Returns latents (B,T,4,h,w) fp32.
    """
    cfg = _cfg(cfg)
    T = cfg["num_frames"]
    B = noise.shape[0]
    sig = karras_sigmas(num_steps)
    ts = sigma_to_timestep(sig)
    if np.asarray(actions).ndim == 3:    # manipulation: continuous (B, T, 8) rows (eval_inference.py:324-331)
        ids = manip_action_ids(actions)
        if cfg.get("action_input_channel") == T + ids.shape[-1] - 1 and cfg.get("action_input_channel") != ids.shape[-1]:
            ids = action_encode_positional(ids)
        act_ids = torch.from_numpy(ids)
    else:
        act_ids = torch.from_numpy(action_ids_idx_encode(actions))
    lat = noise.float()
    if cfg.get("task_type", "navigation") == "navigation":     # pipeline:352-356: the rotation is navigation-only
        lat = rotate_latent_noise(lat, decode_action_seq_frames(act_ids.numpy()))
    lat = lat * init_noise_sigma(sig)
    g = torch.linspace(min_guidance, max_guidance, T).reshape(1, T, 1, 1, 1)  # pipeline:576-581
    tid = torch.tensor([[fps - 1, motion_bucket_id, noise_aug_strength]], dtype=torch.float32)  # pipeline:518, 254-280
    traj = []
    out = []
    for b in range(B):
        x = lat[b:b + 1]
        il = torch.cat([torch.zeros_like(image_latents[b:b + 1]), image_latents[b:b + 1]]).float()
        il = il[:, None].repeat(1, T, 1, 1, 1)  # pipeline:543
        ehs = torch.cat([torch.zeros_like(image_embeddings[b:b + 1]), image_embeddings[b:b + 1]]).float()
        for i in range(num_steps):
            s, sn = float(sig[i]), float(sig[i + 1])
            inp = torch.cat([x, x]) / ((s ** 2 + 1) ** 0.5)  # pipeline:589-590, scheduler:313-314
            inp = torch.cat([inp, il], dim=2)  # pipeline:593
            v = unet_forward(sd, cfg, inp, torch.tensor(float(ts[i])), ehs, tid.repeat(2, 1), act_ids[b:b + 1])
            vu, vc = v.chunk(2)
            v = vu + g * (vc - vu)  # pipeline:606-608
            x = euler_step(v, x, s, sn)
            if store_model_dtype is not None:  # scheduler:673 casts back to the model dtype
                x = x.to(store_model_dtype).float()
            if return_trajectory and b == 0:
                traj.append(x.clone())
        out.append(x)
    res = torch.cat(out)
    return (res, traj) if return_trajectory else res


# ----------------------------------------------------------------------------------------------
# response plumbing (the uint8 edges of the worker)
# ----------------------------------------------------------------------------------------------
def frames_to_uint8_response(video: np.ndarray) -> np.ndarray:
    """`process_output_dict` downstream/api_models/__init__.py:226-237: clip to [0,1], *255, TRUNCATE."""
    return (np.clip(video, 0, 1) * 255).astype(np.uint8)


def postprocess_to_uint8_round(frames: np.ndarray) -> np.ndarray:
    """`numpy_to_pil` dp/image_processor.py:133-147: (x*255).round() — frames already in [0,1]."""
    return (frames * 255).round().astype("uint8")
