"""The SVD denoising loop on the HIP UNet — the hot loop of
`StableVideoDiffusionPipeline.__call__` (reference:
FTsvd/diffusers-private/diffusers/pipelines/stable_video_diffusion/pipeline_stable_video_diffusion.py:555-622)
with the EulerDiscreteScheduler of the served config
(FTsvd/diffusers-private/diffusers/schedulers/scheduling_euler_discrete.py:319-450, 576-684).

Latents stay fp32 in HBM for the whole loop (the reference stores them in the model dtype between
steps, scheduler:673; fp32 is strictly more accurate).  Per step: one `wiw_prep_unet_input`
(scale_model_input + CFG doubling + concat), one UNet forward, one `wiw_cfg_euler_step`.
Candidates are batched: candidate i of a batch equals the reference's B=1 run on candidate i.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import numpy as np
import torch

from .config import SchedulerConfig
from .unet import CIN_PAD, UNetHIP

ACTION_STOP = 4  # utils/svd_utils.py:612 action_mapping


def karras_sigmas(num_steps: int, sc: SchedulerConfig = SchedulerConfig()) -> np.ndarray:
    """Karras rho-schedule: float64 ramp -> float32 table, then a trailing 0 (scheduler:476-499, 438-450)."""
    ramp = np.linspace(0, 1, num_steps)
    lo, hi = sc.sigma_min ** (1 / sc.rho), sc.sigma_max ** (1 / sc.rho)
    sig = (hi + ramp * (lo - hi)) ** sc.rho
    return np.concatenate([sig.astype(np.float32), np.zeros(1, np.float32)])


def sigma_to_timestep(sigmas: np.ndarray) -> np.ndarray:
    """Continuous timesteps t = 0.25 ln(sigma) (scheduler:443-444)."""
    return np.array([0.25 * math.log(float(s)) for s in sigmas[:-1]], dtype=np.float32)


def init_noise_sigma(sigmas: np.ndarray) -> float:
    """sqrt(sigma_max^2 + 1) for timestep_spacing='leading' (scheduler:262-268)."""
    return float((float(sigmas.max()) ** 2 + 1) ** 0.5)


def action_ids_idx_encode(actions: np.ndarray) -> np.ndarray:
    """Cumulative action matrix of `utils/svd_utils.py:594-632`: row i = [stop, a1..ai, 0...]."""
    actions = np.asarray(actions)
    B, L = actions.shape
    a = actions.astype(np.float32).copy()
    a[:, 0] = ACTION_STOP
    tri = np.tril(np.ones((L, L), np.float32))
    return a[:, None, :] * tri[None]


def rotate_latent_noise(noise: torch.Tensor, actions: np.ndarray) -> torch.Tensor:
    """`sample_latent_noise` after its randn draw (pipeline:750-786): a turn at frame i makes frame i a
    +-W/16 cyclic shift of frame i-1 (22.5 degrees of the panorama)."""
    W = noise.shape[-1]
    if W % 16:
        raise ValueError("latent width must be divisible by 16 for the 22.5-degree rotation (pipeline:816-819)")
    noise = noise.clone()
    acts = np.asarray(actions)
    for b in range(acts.shape[0]):
        for i in range(1, acts.shape[1]):
            if acts[b, i] == 2:
                noise[b, i] = torch.roll(noise[b, i - 1], W // 16, dims=-1)
            elif acts[b, i] == 3:
                noise[b, i] = torch.roll(noise[b, i - 1], -(W // 16), dims=-1)
    return noise


class SVDDenoiser:
    def __init__(self, unet: UNetHIP, sched: SchedulerConfig = SchedulerConfig()):
        self.unet = unet
        self.hip = unet.hip
        self.sched = sched
        self.device = unet.device

    @torch.no_grad()
    def denoise(self, image_latents: torch.Tensor, image_embeddings: torch.Tensor, noise: torch.Tensor,
                actions: np.ndarray, num_steps: int = 25, fps: int = 7, motion_bucket_id: int = 127,
                noise_aug_strength: float = 0.02, min_guidance: float = 1.0, max_guidance: float = 3.0,
                callback=None) -> torch.Tensor:
        """image_latents (B,4,h,w): VAE mode() of the noise-augmented conditioning image;
        image_embeddings (B,1,D): CLIP image embeds; noise (B,T,4,h,w): unit Gaussian draw (before the
        action rotation and init_noise_sigma); actions (B,T) ints.  Returns latents (B,T,4,h,w) fp32.
        fps / motion_bucket_id are accepted for API parity: `aug_emb` is dead for micro_cond (unet:482)."""
        cfg = self.unet.cfg
        B, T, _, h, w = noise.shape
        assert T == cfg.num_frames and image_latents.shape == (B, 4, h, w)
        sig = karras_sigmas(num_steps, self.sched)
        ts = sigma_to_timestep(sig)
        act_ids = action_ids_idx_encode(actions)
        decoded = np.diagonal(act_ids, axis1=-2, axis2=-1)  # decode_action_seq_frames (pipeline:826-844)
        lat = rotate_latent_noise(noise.to(self.device, torch.float32), decoded) * init_noise_sigma(sig)
        lat = lat.contiguous()
        img = image_latents.to(self.device, torch.float32).contiguous()
        cond = self.unet.prepare_request(image_embeddings, act_ids, noise_aug_strength)
        hw = h * w
        x_in = torch.empty((2 * B * T * hw, CIN_PAD), dtype=self.unet.dtype, device=self.device)
        for i in range(num_steps):
            s, sn = float(sig[i]), float(sig[i + 1])
            self.hip.prep_unet_input(lat, img, B, T, hw, s, CIN_PAD, x_in)
            emb = self.unet.time_embedding(float(ts[i]), cond)
            v = self.unet.forward(x_in, emb, cond, h, w)
            self.hip.cfg_euler_step(v, cfg.out_channels, lat, B, T, hw, s, sn, min_guidance, max_guidance)
            if callback is not None:
                callback(i, lat)
        return lat
