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
import os
import time
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


class GraphedForward:
    """ONE captured UNet forward (hipGraph through torch.cuda.CUDAGraph) for a (candidates, h, w) shape: ~1 100 kernel launches
    — ~20 ms of host time per forward when issued one by one through ctypes — become one `hipGraphLaunch`.  The C ABI
    allocates and synchronises nothing (include/wiw_svd.h), so the launches are capture-safe as they are; the intermediates
    of the forward live in the graph's private pool.  Inputs and output are STATIC buffers: `x_in` (written by
    wiw_prep_unet_input), `emb` (the per-step time embedding), the request's conditioning (`cond`: copied in by `load`),
    `v` (read by wiw_cfg_euler_step).  Replay is bit-identical to the eager forward (same kernels, same arguments)."""

    def __init__(self, unet: UNetHIP, cond, h: int, w: int):
        cfg = unet.cfg
        self.unet, self.h, self.w, self.cond = unet, h, w, cond            # `cond`'s tensors become the static ones
        rows = cond.Bc * cfg.num_frames * h * w
        self.x_in = torch.zeros((rows, CIN_PAD), dtype=unet.dtype, device=unet.device)
        self.emb = torch.zeros((cond.Bc * cfg.num_frames, cfg.time_embed_dim), dtype=unet.dtype, device=unet.device)
        hip = unet.hip
        assert hip.gemm_profile is None and hip.kernel_profile is None, "per-launch events cannot be recorded inside a capture"
        side = torch.cuda.Stream(unet.device)
        side.wait_stream(torch.cuda.current_stream(unet.device))
        with torch.cuda.stream(side):                                       # warm-up: one-time attribute calls, workspaces
            unet.forward(self.x_in, self.emb, cond, h, w)
        torch.cuda.current_stream(unet.device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        # thread-local capture mode: with a process group alive, RCCL's watchdog thread polls events from ANOTHER thread; in the
        # default (global) mode such a call during the capture would invalidate it.  This thread only launches kernels.
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.v = unet.forward(self.x_in, self.emb, cond, h, w)

    def load(self, cond) -> None:
        """Copy a new request's conditioning (same candidate count) into the static tensors."""
        dst = self.cond
        assert (cond.B, cond.Bc) == (dst.B, dst.Bc)
        if cond is dst:
            return
        for name in ("act_emb", "noise_emb", "ehs_bf16"):
            getattr(dst, name).copy_(getattr(cond, name))
        for name in ("cross", "pos_emb", "pos_emb_blend"):
            for k, t in getattr(cond, name).items():
                getattr(dst, name)[k].copy_(t)

    def __call__(self, emb: torch.Tensor) -> torch.Tensor:
        self.emb.copy_(emb)
        self.graph.replay()
        return self.v


class SVDDenoiser:
    def __init__(self, unet: UNetHIP, sched: SchedulerConfig = SchedulerConfig(), use_graph: Optional[bool] = None):
        """use_graph: replay the UNet forward from a captured hipGraph (`GraphedForward`, one per (candidates, h, w));
        default: env WIW_GRAPH=1.  Same bytes as the eager path; removes ~20 ms of host launch work per forward — the GPU
        side of a 576x1024 forward is ~100 ms, so it matters for small latents and for a host that is also serving."""
        self.unet = unet
        self.hip = unet.hip
        self.sched = sched
        self.device = unet.device
        self.use_graph = bool(os.environ.get("WIW_GRAPH")) if use_graph is None else bool(use_graph)
        self._graphs = {}           # (B, h, w) -> GraphedForward; at most MAX_GRAPHS shapes stay captured (their pools hold a
        self.MAX_GRAPHS = 2         # forward's intermediates: ~6 GB per candidate at 576x1024)
        self.graph_error = None     # why capture was given up (the loop then runs eagerly)
        # host time spent ENQUEUEING UNet forwards (no synchronisation inside): [seconds, forwards] per mode
        self.host_launch = {"eager": [0.0, 0], "graph": [0.0, 0]}

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
        gf = None
        if self.use_graph and self.graph_error is None:
            gf = self._graphs.pop((B, h, w), None)
            if gf is not None:
                self._graphs[(B, h, w)] = gf     # re-insert: dict order = recency, eviction below is least-recently-used
            if gf is None:
                prof = (self.hip.gemm_profile, self.hip.kernel_profile)      # (bench.py may have armed per-launch events for
                self.hip.gemm_profile = self.hip.kernel_profile = None       # the first step: not inside a capture)
                try:
                    while len(self._graphs) >= self.MAX_GRAPHS:
                        self._graphs.pop(next(iter(self._graphs)))
                    gf = self._graphs[(B, h, w)] = GraphedForward(self.unet, cond, h, w)
                except Exception as e:       # capture is an optimisation: the eager loop below is the same computation
                    self.graph_error = f"{type(e).__name__}: {e}"
                    gf = None
                finally:
                    self.hip.gemm_profile, self.hip.kernel_profile = prof
            if gf is not None:
                gf.load(cond)
                cond = gf.cond
        x_in = gf.x_in if gf is not None else torch.empty((2 * B * T * hw, CIN_PAD), dtype=self.unet.dtype, device=self.device)
        for i in range(num_steps):
            s, sn = float(sig[i]), float(sig[i + 1])
            self.hip.prep_unet_input(lat, img, B, T, hw, s, CIN_PAD, x_in)
            emb = self.unet.time_embedding(float(ts[i]), cond)
            # (a step that carries per-launch events — bench.py — runs eagerly on the same static buffers)
            replay = gf is not None and self.hip.gemm_profile is None and self.hip.kernel_profile is None
            t0 = time.perf_counter()
            v = gf(emb) if replay else self.unet.forward(x_in, emb, cond, h, w)
            hl = self.host_launch["graph" if replay else "eager"]
            hl[0] += time.perf_counter() - t0
            hl[1] += 1
            self.hip.cfg_euler_step(v, cfg.out_channels, lat, B, T, hw, s, sn, min_guidance, max_guidance)
            if callback is not None:
                callback(i, lat)
        return lat
