"""Fine-tuning step of the action-conditioned SVD UNet — SURVEY.md 8(f) row 2 (`FTsvd/train_svd.py:844-970`).

This module holds the pieces of the step that are not the network itself: the host-side preparation (EDM noise-level
draws, pre-conditioning of the UNet input, conditioning dropout — O(latent) elementwise work on 57 k-element tensors,
PyTorch as plumbing), the EDM loss with its gradient (`TrainStep.loss_and_grad`, `wiw_edm_loss_grad`), the building blocks
of the backward pass of a linear / convolution layer (`linear_backward`, `conv_backward`: data gradients on `wiw_gemm_bf16`,
weight gradients through the measured / modelled plans of `wgrad_gemm` — `wiw_wgrad_tn_bf16` or the GEMM on transposed
operands), and the learning-rate schedules (`lr_at`).  The network's training forward / backward (every operator of the UNet
on the HIP kernels of `csrc/train.hip`, gradients pinned to the reference's `loss.backward()`) is `train_unet.UNetTrain`;
`train_unet.Trainer` is the step (forward, loss, backward, AdamW, ZeRO-1 reduction through `parallel.ShardedAdamW`,
accumulation, loss scaling, checkpoints).  Checker of the whole step: `oracle/train_oracle.py`, pinned to the reference's
autograd (`tests/golden/train_step_tiny*.npz`).  DESIGN.md 8.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from .hip import Hip


def rand_log_normal(shape, loc: float = 0.0, scale: float = 1.0, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """train_svd.py:69-72: exp(Normal(loc, scale).icdf(u)), u uniform in [1e-7, 1 - 1e-7]; sigma ~ (0.7, 1.6) for the
    diffusion noise (:886), (-3.0, 0.5) for the noise on the conditioning image (:876)."""
    u = torch.rand(shape, generator=generator) * (1 - 2e-7) + 1e-7
    return torch.distributions.Normal(loc, scale).icdf(u).exp()


def apply_conditioning_dropout(ehs: torch.Tensor, cond_latents: torch.Tensor, action_ids: torch.Tensor, prob: Optional[float],
                               random_p: torch.Tensor):
    """utils/svd_utils.py:176-241 (micro_cond): image embedding zeroed where p < 2 prob, conditioning latents where
    prob <= p < 3 prob, actions untouched; `random_p` (bsz,) is the uniform draw."""
    if prob is None:
        return ehs, cond_latents, action_ids
    bsz = ehs.shape[0]
    ehs = torch.where((random_p < 2 * prob).reshape(bsz, 1, 1), torch.zeros_like(ehs), ehs)
    keep = 1.0 - ((random_p >= prob).to(cond_latents.dtype) * (random_p < 3 * prob).to(cond_latents.dtype))
    return ehs, keep.reshape(bsz, 1, 1, 1) * cond_latents, action_ids


@dataclass
class StepInputs:
    """What one step feeds the UNet and the loss (one sample per GPU, as the reference: train_svd.py:877)."""
    unet_input: torch.Tensor      # (1,T,8,h,w) fp32: noisy / sqrt(sigma^2 + 1) | conditioning latents
    timestep: float               # 0.25 ln sigma
    noisy: torch.Tensor           # (1,T,4,h,w)
    target: torch.Tensor          # the clean latents
    sigma: float
    added_time_ids: torch.Tensor  # (1,3) = (7, 127, noise_aug_strength)
    ehs: torch.Tensor
    action_ids: torch.Tensor


def prepare_step(latents: torch.Tensor, noise: torch.Tensor, sigma: float, cond_latents: torch.Tensor, ehs: torch.Tensor,
                 noise_aug_strength: float, action_ids: torch.Tensor, dropout_prob: Optional[float] = None,
                 random_p: Optional[torch.Tensor] = None) -> StepInputs:
    """train_svd.py:888-931 for one sample: noisy = latents + sigma noise; input = noisy / sqrt(sigma^2 + 1) concatenated
    with the (possibly dropped) conditioning latents on the channel axis; t = 0.25 ln sigma."""
    assert latents.shape[0] == 1 and latents.shape == noise.shape
    ehs, cond_latents, action_ids = apply_conditioning_dropout(ehs, cond_latents, action_ids, dropout_prob, random_p)
    noisy = latents + noise * sigma
    inp = torch.cat([noisy / (sigma ** 2 + 1) ** 0.5, cond_latents.unsqueeze(1).repeat(1, latents.shape[1], 1, 1, 1)], dim=2)
    tids = torch.tensor([[7.0, 127.0, float(noise_aug_strength)]])
    return StepInputs(inp, 0.25 * float(torch.log(torch.tensor(float(sigma)))), noisy, latents, float(sigma), tids, ehs, action_ids)


@torch.no_grad()
def batch_from_pixels(front, pixel_values: torch.Tensor, past_obs: torch.Tensor, actions, *, scaling_factor: Optional[float] = None,
                      dropout_prob: Optional[float] = None, generator: Optional[torch.Generator] = None,
                      draws: Optional[dict] = None) -> StepInputs:
    """One training sample from PIXELS, as the reference's loop builds it (train_svd.py:846-931), on the HIP front end:
        latents      = vae.encode(frames).latent_dist.sample() * scaling_factor            (:86-95, :856)
        noise        = sample_latent_noise(actions, ...)                                    (:873; pipeline:750-786)
        cond_sigma   = rand_log_normal(-3.0, 0.5); cond image = frame 0 + randn * cond_sigma   (:876-880)
        cond_latents = vae.encode(cond image).latent_dist.sample()   [* scaling / scaling]   (:881-882)
        sigma        = rand_log_normal(0.7, 1.6)                                             (:886)
        ehs          = CLIP(resize_224(past_obs))   (norm_image + feature_extractor normalise + image_encoder, :893-894)
    then `prepare_step` (noisy latents, EDM input scaling, conditioning dropout, t = 0.25 ln sigma).
    front: `vae.HIPFrontend` (VAEHIP.encode_moments, CLIPVisionHIP).  pixel_values (1, T, 3, H, W) and past_obs (1, P, 3, H', W')
    fp32 in [-1, 1]; actions (1, T) ints.  `draws` overrides the random draws (tests / replay): vae_eps (T, L, h, w), cond_eps
    (1, L, h, w), pixel_noise (1, 3, H, W), latent_noise (1, T, L, h, w), sigma, cond_sigma (floats), random_p ((1,))."""
    import numpy as np

    from .pipeline import action_ids_idx_encode, rotate_latent_noise

    d = dict(draws or {})
    vae = front.vae
    sf = float(vae.scaling_factor if scaling_factor is None else scaling_factor)
    assert pixel_values.dim() == 5 and pixel_values.shape[0] == 1, "one sample per step, as the reference (train_svd.py:877)"
    T = pixel_values.shape[1]
    dev = vae.device
    px = pixel_values[0].to(dev, torch.float32)

    def randn(name, shape):
        if name in d:
            return torch.as_tensor(d[name], dtype=torch.float32).to(dev)
        return torch.randn(shape, generator=generator).to(dev)

    def sample(x, eps_name):                                   # DiagonalGaussianDistribution.sample(): mean + std * eps
        mean, logvar = vae.encode_moments(x)
        return mean + torch.exp(0.5 * logvar) * randn(eps_name, tuple(mean.shape))

    latents = (sample(px, "vae_eps") * sf)[None]                                             # (1, T, L, h, w)
    acts = np.asarray(actions).reshape(1, T)
    noise = rotate_latent_noise(randn("latent_noise", tuple(latents.shape)), acts)
    cond_sigma = float(d["cond_sigma"]) if "cond_sigma" in d else float(rand_log_normal([1], -3.0, 0.5, generator)[0])
    cond_px = px[0:1] + randn("pixel_noise", tuple(px[0:1].shape)) * cond_sigma
    cond_latents = sample(cond_px, "cond_eps")                                                # * sf / sf
    sigma = float(d["sigma"]) if "sigma" in d else float(rand_log_normal([1], 0.7, 1.6, generator)[0])
    po = past_obs.to(dev, torch.float32)
    assert po.reshape(-1, *po.shape[-3:]).shape[0] == 1, "one past observation per sample (train_svd.sh; UNetTrain takes (1, 1, D))"
    e = front._embed(po.reshape(-1, *po.shape[-3:]), None)                                    # (1, 1, D) fp32
    ehs = e.reshape(1, 1, e.shape[-1])
    action_ids = torch.from_numpy(action_ids_idx_encode(acts))
    random_p = None
    if dropout_prob is not None:
        random_p = torch.as_tensor(d["random_p"]) if "random_p" in d else torch.rand(1, generator=generator)
    return prepare_step(latents.cpu(), noise.cpu(), sigma, cond_latents.cpu(), ehs.cpu(), cond_sigma, action_ids, dropout_prob, random_p)


# ------------------------------------------------------------------------------------------------
# backward building blocks on the existing kernels (no new GEMM code: the gradients of a GEMM are GEMMs on transposed operands)
# ------------------------------------------------------------------------------------------------
def lr_at(step: int, base_lr: float, schedule: str = "cosine", warmup_steps: int = 500, total_steps: int = 1002) -> float:
    """Learning rate of optimiser step `step` (0-based) under the reference's `get_scheduler(args.lr_scheduler, ...)`
    (train_svd.py:1131-1136; `diffusers.optimization`, a LambdaLR on the base rate; train_svd.sh runs "cosine", warm-up 500
    (:237-240), 1002 steps).  The published multipliers:
        constant               1
        constant_with_warmup   min(1, step / warmup)
        linear                 step / warmup during warm-up, then max(0, (total - step) / (total - warmup))
        cosine                 step / warmup during warm-up, then max(0, 0.5 (1 + cos(pi * progress))), progress =
                               (step - warmup) / max(1, total - warmup)   (num_cycles = 0.5)"""
    import math

    w = max(1, warmup_steps)
    if schedule == "constant":
        return base_lr
    if step < warmup_steps:
        return base_lr * step / w
    if schedule == "constant_with_warmup":
        return base_lr
    if schedule == "linear":
        return base_lr * max(0.0, (total_steps - step) / max(1, total_steps - warmup_steps))
    if schedule == "cosine":
        progress = (step - warmup_steps) / max(1, total_steps - warmup_steps)
        return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * progress)))
    raise ValueError(f"unknown lr schedule {schedule!r}")


def wgrad_plan(n_out: int, k_in: int, k_rows: int, n_cu: int = 256):
    """(flip, splitk) of a weight-gradient GEMM dW [n_out, k_in] = dy^T x whose K loop runs over k_rows (the M rows of the
    layer: up to 129 024) while the output is a handful of tiles (dW of a 320 x 320 projection: 2 x 2 tiles of 256 x 160 for
    256 CUs).  Two choices, priced by a small model of `wiw_gemm_bf16`'s persistent schedule (one block per CU):
      * orientation: dW (GEMM rows = n_out) or dW^T (flip: GEMM rows = k_in) — rows come in 256-row tiles, so the 320 outputs
        of a level-0 projection waste 37 % of two row tiles while its 1280 / 2880 inputs fill theirs;
      * split-K factor S (WiwGemmArgs.splitk: fp32 slabs + a deterministic reduce), a divisor of the K-tile count with >= 8
        K tiles per range: the items (tiles x S) run in ceil(items / CUs) rounds, so 256 items beat 288; the launch takes
        the 256 x 320 tile under the same rule as gemm.hip (`launch`), which halves the item count.
    cost [K-tile units of the 256 x 160 tile] = rounds x (K tiles per range x tile area / efficiency + fixed per item)
    + slab reduce.  Calibrated on the served shapes (`tools/train_probe.py`, gpurun_out r03q: the plan is within a few
    percent of the best of an exhaustive sweep; the previous fixed rule was 1.0-1.9x off)."""
    nk = k_rows // 64
    best = None
    for flip in (0, 1):
        gm, gn = (k_in, n_out) if flip else (n_out, k_in)
        mt = -(-gm // 256)
        for sk in range(1, 129):
            if nk % sk or (sk > 1 and nk // sk < 8):
                continue
            kt = nk // sk
            items_h = mt * -(-gn // 320) * sk
            huge = gn % 320 == 0 and kt * 64 >= 640 and (items_h >= 200 or sk == 1)
            items = items_h if huge else mt * -(-gn // 160) * sk
            rounds = -(-items // n_cu)
            per_item = kt * 1.8 + 16.0 if huge else kt + 10.0
            cost = rounds * per_item + (sk * gm * gn / 1.2e6 if sk > 1 else 0.0)
            if best is None or cost < best[0] - 1e-9:
                best = (cost, flip, sk)
    return best[1], best[2]


# Measured plans (key "n_out,k_in,rows[,c][,v]" -> [mode, splits]; mode 0 / 1 = `wiw_gemm_bf16` on transposed operands, dW /
# dW^T; mode 2 = `wiw_wgrad_tn_bf16` on the row-major operands, mode 3 = the same with the operands swapped, dW^T).  The weight-gradient GEMMs stream BOTH operands over a K loop of
# up to 129 024 rows, so their speed is decided by how the concurrently running items share operand panels in the per-XCD L2s
# — which the schedule model of `wgrad_plan` does not see (it is 1.0-1.7x off on the convolutions).  With tuning on
# (`set_wgrad_tuning(True)`, `Trainer(autotune=True)`, `bench.py --train`), the first call of a shape times every candidate
# with HIP events and keeps the fastest — what MIOpen's find step does for its convolutions.  A plan never changes inside a
# process once chosen, and `Trainer.save` / `load` carry the table, so a resumed run repeats the same summation orders.
_WGRAD_PLANS: dict = {}
_WGRAD_TUNE = False


def set_wgrad_tuning(on: bool) -> None:
    global _WGRAD_TUNE
    _WGRAD_TUNE = bool(on)


def clear_wgrad_plans() -> None:
    _WGRAD_PLANS.clear()


def wgrad_plans() -> dict:
    return dict(_WGRAD_PLANS)


def load_wgrad_plans(plans: dict) -> None:
    _WGRAD_PLANS.update({k: (int(v[0]), int(v[1])) for k, v in plans.items()})


def _wgrad_run(hip: Hip, dyT, xT, n_out, k_in, k_rows, flip, sk, view_ok=False):
    from .hip import EPI_OUT_F32

    if flip:
        out = torch.empty(k_in, n_out, dtype=torch.float32, device=hip.device)
        hip.gemm(xT, dyT, out, M=k_in, N=n_out, K=k_rows, C1=k_rows, epilogue=EPI_OUT_F32, splitk=sk)
        # the strided copy back is part of a flipped plan's price (and of its timing) unless the caller re-lays the
        # gradient out anyway (the convolutions: (O, ky, kx, I) -> (O, I, ky, kx) is one copy from either orientation)
        return out.t() if view_ok else out.t().contiguous()
    out = torch.empty(n_out, k_in, dtype=torch.float32, device=hip.device)
    hip.gemm(dyT, xT, out, M=n_out, N=k_in, K=k_rows, C1=k_rows, epilogue=EPI_OUT_F32, splitk=sk)
    return out


def wgrad_tn_splits(n_out: int, k_in: int, rows: int, n_cu: int = 256) -> int:
    """Row splits of the row-major weight-gradient kernel (`wiw_wgrad_tn_bf16`, 256 x 128 output tiles, two workgroups
    resident per CU): about 480 (tile, split) items — one full set of resident workgroups — with at least 128 rows per split
    (sweep of the served shapes, profiles/r04d_wgrad_tn_sweep.txt)."""
    tiles = -(-n_out // 256) * -(-k_in // 128)
    return int(max(1, min(round(1.875 * n_cu / tiles), rows // 128, 512)))


def wgrad_default_plan(n_out: int, k_in: int, rows: int, conv: bool):
    """(mode, splits) without measuring.  mode 2 / 3 = `wiw_wgrad_tn_bf16` on the row-major operands (no transposes): ahead on
    every linear layer of the served architecture once the two transposes the other modes need are counted; mode 0 / 1 =
    `wiw_gemm_bf16` on transposed operands (dW / dW^T, `wgrad_plan`): still ahead on the long-K convolutions, whose
    operand arrives transposed from `wiw_gather_taps_t_bf16` anyway."""
    if not conv:
        # the kernel's tile is 256 (first operand's columns) x 128: the orientation with the smaller padded output (mode 3
        # computes dW^T with the operands swapped: a 320-wide dy fills 3 x 128 better than 2 x 256)
        pad = lambda a, b: -(-a // 256) * 256 * (-(-b // 128) * 128)  # noqa: E731
        if 1.25 * pad(k_in, n_out) < pad(n_out, k_in):                # (the copy back has to be paid for: measured plans agree)
            return 3, wgrad_tn_splits(k_in, n_out, rows)
        return 2, wgrad_tn_splits(n_out, k_in, rows)
    return wgrad_plan(n_out, k_in, -(-rows // 64) * 64)


def _wgrad_operands_nt(hip: Hip, dy, x, M, N, K, conv):
    """K-contiguous operands of the `wiw_gemm_bf16` modes: dy^T and x^T (or the transposed im2col rows), rows zero-padded to 64."""
    Mp = -(-M // 64) * 64
    alloc = torch.empty if Mp == M else torch.zeros
    dyT = alloc(N, Mp, dtype=hip.dtype, device=hip.device)
    hip.transpose(dy, dy.stride(0), 0, M, N, dyT, Mp)
    if conv is None:
        xT = alloc(K, Mp, dtype=hip.dtype, device=hip.device)
        hip.transpose(x, x.stride(0), 0, M, K, xT, Mp)
    else:
        xT = hip.gather_taps_t(x, M, *conv)                          # [taps * Cin, Mp], one pass over x
    return dyT, xT, Mp


def _wgrad_exec(hip: Hip, plan, dy, x, M, N, K, conv, view_ok):
    if plan[0] >= 2:
        xr = x if conv is None else hip.gather_taps(x, M, *conv)      # row-major im2col rows [M, taps * Cin]
        if plan[0] == 2:
            return hip.wgrad_tn(dy, xr, M, N, K, plan[1])
        out = hip.wgrad_tn(xr, dy, M, K, N, plan[1]).t()               # mode 3: dW^T = X^T dY, operands swapped
        return out if view_ok else out.contiguous()
    dyT, xT, Mp = _wgrad_operands_nt(hip, dy, x, M, N, K, conv)
    return _wgrad_run(hip, dyT, xT, N, K, Mp, plan[0], plan[1], view_ok)


def _time(fn, reps=2):
    fn()                                                              # warm (workspace allocation, code load)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps


def _wgrad_tune(hip: Hip, dy, x, M, N, K, conv, view_ok):
    """Every candidate once, end to end: the row-major kernel over its row splits; the `wiw_gemm_bf16` modes over orientation
    and split-K with the time of making their transposed operands added.  Ties go to the default plan."""
    default = wgrad_default_plan(N, K, M, conv is not None)
    cands = []
    xr = x if conv is None else hip.gather_taps(x, M, *conv)
    t_gather = 0.0 if conv is None else _time(lambda: hip.gather_taps(x, M, *conv))
    for mode in (2, 3):
        s0 = wgrad_tn_splits(N, K, M) if mode == 2 else wgrad_tn_splits(K, N, M)
        for sp in sorted({1, max(1, s0 // 4), max(1, s0 // 2), max(1, (3 * s0) // 4), s0, (3 * s0) // 2, 2 * s0, 3 * s0}):
            if sp >= 1 and M // sp >= 64 and sp * N * K * 4 <= (1 << 30):
                cands.append((_time(lambda: _wgrad_exec(hip, (mode, sp), dy, xr, M, N, K, None, view_ok)) + t_gather, (mode, sp)))
    del xr
    dyT, xT, Mp = _wgrad_operands_nt(hip, dy, x, M, N, K, conv)
    t_prep = _time(lambda: _wgrad_operands_nt(hip, dy, x, M, N, K, conv))
    nk = Mp // 64
    for f in (0, 1):
        for sk in range(1, 129):
            if nk % sk == 0 and (sk == 1 or nk // sk >= 8) and sk * N * K * 4 <= (1 << 30):
                cands.append((_time(lambda: _wgrad_run(hip, dyT, xT, N, K, Mp, f, sk, view_ok)) + t_prep, (f, sk)))
    best = min(cands, key=lambda c: c[0] * (1.0 if c[1] == default else 1.03))
    return best[1]


def wgrad(hip: Hip, dy: torch.Tensor, x: torch.Tensor, M: int, N: int, K: int, view_ok: bool = False, conv=None) -> torch.Tensor:
    """fp32 dW [N, K] = dy[:M]^T . X[:M] with the plan of the shape.  dy [>= M rows, N] row-major; X = x [>= M rows, K], or,
    conv = (Cin, H, W, T, temporal, stride), the im2col rows of the activation x (K = taps * Cin), built in whichever layout
    the plan's kernel reads.  view_ok: a flipped `wiw_gemm_bf16` plan may return the transposed view (the caller re-lays
    the gradient out anyway)."""
    key = f"{N},{K},{M}" + (",c" if conv is not None else "") + (",v" if view_ok else "")
    plan = _WGRAD_PLANS.get(key)
    if plan is None:
        plan = _wgrad_tune(hip, dy, x, M, N, K, conv, view_ok) if _WGRAD_TUNE else wgrad_default_plan(N, K, M, conv is not None)
        _WGRAD_PLANS[key] = plan
    return _wgrad_exec(hip, plan, dy, x, M, N, K, conv, view_ok)


def wgrad_gemm(hip: Hip, dyT: torch.Tensor, xT: torch.Tensor, n_out: int, k_in: int, k_rows: int, view_ok: bool = False) -> torch.Tensor:
    """The `wiw_gemm_bf16` form alone, on operands that are ALREADY K-contiguous (dyT [n_out, k_rows], xT [k_in, k_rows]):
    orientation / split-K from the schedule model."""
    flip, sk = wgrad_plan(n_out, k_in, k_rows)
    return _wgrad_run(hip, dyT, xT, n_out, k_in, k_rows, flip, sk, view_ok)


def linear_backward(hip: Hip, x: torch.Tensor, W: torch.Tensor, dy: torch.Tensor, need_dx: bool = True, need_db: bool = True,
                    need_dw: bool = True):
    """Backward of y = x . W^T (+ b) with x [M, K], W [N, K], dy [M, N] in the Hip's 16-bit type (M, N, K % 64 == 0):
         dx [M, K] (16-bit)  = dy . W              -> wiw_gemm_bf16(A = dy, W = W^T)
         dW [N, K] (fp32)    = dy^T . x            -> `wgrad`: wiw_wgrad_tn_bf16 on the row-major operands (or wiw_gemm_bf16 on
                                                      transposed ones), contraction over the M rows
         db [N]   (fp32)     = column sums of dy   -> wiw_colsum
       (`nn.Linear` backward, attention_processor.py:2358-2391 / attention.py:1185-1243 call sites).  The transposes are
       wiw_transpose_bf16 passes; fp32 accumulation over all M rows inside the MFMA K loop."""
    M, K = x.shape
    N = W.shape[0]
    assert W.shape == (N, K) and dy.shape == (M, N) and M % 8 == 0 and N % 64 == 0 and K % 64 == 0
    dev, dt = hip.device, hip.dtype
    dx = None
    if need_dx:
        Wt = torch.empty(K, N, dtype=dt, device=dev)
        hip.transpose(W, K, 0, N, K, Wt, N)
        dx = torch.empty(M, K, dtype=dt, device=dev)
        hip.gemm(dy, Wt, dx, M=M, N=K, K=N, C1=N)
    dW = None
    if need_dw:                        # frozen weights (`--train_param_type new`): no weight-gradient GEMM
        dW = wgrad(hip, dy, x, M, N, K)
    db = hip.colsum(dy, M, N) if need_db else None
    return dx, dW, db


def conv_backward(hip: Hip, x: torch.Tensor, Wk: torch.Tensor, dy: torch.Tensor, H: int, Wd: int, T: int = 1,
                  temporal: bool = False, need_dx: bool = True, need_dw: bool = True):
    """Backward of the stride-1 implicit-GEMM convolutions of `wiw_gemm_bf16`: 3x3 pad 1 (ResnetBlock2D convs, resnet.py:269,285)
    or, temporal=True, (3,1,1) pad 1 over T (TemporalResnetBlock, resnet.py:570-592).  x [M, Cin] token-major, Wk [Cout, taps*Cin]
    in the kernel's layout ([Cout][ky][kx][Cin] / [Cout][kt][Cin]), dy [M, Cout]; Cin, Cout, M % 64 == 0.
        dx  = the SAME convolution of dy with the taps mirrored and the channel roles swapped (W2[ci][tap'][co] = W[co][tap][ci])
        dW  = dy^T . im2col(x)   (fp32 [Cout, taps*Cin]: `wgrad` with the im2col operand built in the layout its plan reads)
        db  = column sums of dy
    (the stride-2 / upsampling variants are not covered yet)."""
    from .hip import A_CONV3X3, A_CONV_T3

    M, Cin = x.shape
    Cout = Wk.shape[0]
    taps = 3 if temporal else 9
    assert Wk.shape == (Cout, taps * Cin) and dy.shape == (M, Cout) and M % 8 == 0 and Cin % 64 == 0 and Cout % 64 == 0
    dev, dt = hip.device, hip.dtype
    dx = None
    if need_dx:
        W2 = Wk.reshape(Cout, taps, Cin).flip(1).permute(2, 1, 0).reshape(Cin, taps * Cout).contiguous()   # host re-layout
        dx = torch.empty(M, Cin, dtype=dt, device=dev)
        hip.gemm(dy, W2, dx, M=M, N=Cin, K=taps * Cout, C1=Cout, mode=A_CONV_T3 if temporal else A_CONV3X3, H=H, Wd=Wd, T=T)
    if not need_dw:
        return dx, None, None
    dW = wgrad(hip, dy, x, M, Cout, taps * Cin, view_ok=True, conv=(Cin, H, Wd, T, temporal, 1))   # possibly a transposed view
    return dx, dW, hip.colsum(dy, M, Cout)


class FeedForwardTrain:
    """Forward-with-saved-activations and backward of ONE residual FeedForward sub-block,
        y = x + W2 . GEGLU(W1 . LayerNorm(x) + b1) + b2        (BasicTransformerBlock norm3 + ff, attention.py:565-582),
    composed from the building blocks above: the template for the full-network orchestration that does not exist yet.
    Weights are 16-bit copies [N, K] (natural row order, no GEGLU packing), gradients fp32."""

    def __init__(self, hip: Hip, gamma, beta, W1, b1, W2, b2, eps: float = 1e-5):
        self.hip, self.eps = hip, eps
        self.gamma, self.beta, self.W1, self.b1, self.W2, self.b2 = gamma, beta, W1, b1, W2, b2
        self.C = W1.shape[1]
        assert W1.shape == (8 * self.C, self.C) and W2.shape == (self.C, 4 * self.C)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        hip, C = self.hip, self.C
        M = x.shape[0]
        self.x = x
        self.a = hip.layernorm(x, M, C, self.gamma, self.beta, self.eps)
        self.P = torch.empty(M, 8 * C, dtype=hip.dtype, device=hip.device)
        hip.gemm(self.a, self.W1, self.P, M=M, N=8 * C, K=C, C1=C, bias=self.b1)          # saved for the GEGLU backward
        self.h = hip.geglu_fwd(self.P, M, 4 * C)
        y = torch.empty(M, C, dtype=hip.dtype, device=hip.device)
        hip.gemm(self.h, self.W2, y, M=M, N=C, K=4 * C, C1=4 * C, bias=self.b2, res1=x, ldr1=C, beta1=1.0)
        return y

    def backward(self, dy: torch.Tensor):
        """-> (dx, {name: fp32 gradient})."""
        hip, C = self.hip, self.C
        M = dy.shape[0]
        dh, dW2, db2 = linear_backward(hip, self.h, self.W2, dy)
        dP = hip.geglu_bwd(self.P, dh, M, 4 * C)
        da, dW1, db1 = linear_backward(hip, self.a, self.W1, dP)
        dx, dgamma, dbeta = hip.layernorm_bwd(self.x, da, self.gamma, M, C, self.eps, dres=dy)   # residual join inside the kernel
        return dx, dict(gamma=dgamma, beta=dbeta, W1=dW1, b1=db1, W2=dW2, b2=db2)


class TrainStep:
    """Loss / optimiser side of the step on the HIP kernels; the UNet backward between them is not built yet."""

    def __init__(self, hip: Hip):
        self.hip = hip

    def loss_and_grad(self, model_pred: torch.Tensor, step: StepInputs) -> Tuple[torch.Tensor, torch.Tensor]:
        """EDM loss (train_svd.py:940-952) and dL/d(model_pred), fp32, on the device."""
        dev = self.hip.device
        f = lambda t: t.to(dev, torch.float32).contiguous()  # noqa: E731
        return self.hip.edm_loss_grad(f(model_pred), f(step.noisy), f(step.target), step.sigma)

    def backward(self, *_a, **_k):
        raise NotImplementedError("the backward kernels of the UNet operators are not built yet (SURVEY.md 8(f) row 2)")

    def adamw(self, p32, g32, m, v, step, lr, p16=None, **kw):
        self.hip.adamw_step(p32, g32, m, v, step, lr, p16=p16, **kw)
