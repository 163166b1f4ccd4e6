"""CPU restatement of ONE fine-tuning step of the action-conditioned SVD UNet (BASELINE config 4).

TEST INFRASTRUCTURE ONLY (the oracle of row f2): plain torch fp32 + autograd through `svd_oracle.unet_forward`.  Nothing
under `world-in-world_amd/` imports this module.  Follows the training loop body of the reference,
`/root/reference/FTsvd/train_svd.py:844-970` (EDM pre-conditioning of Karras et al. as used by SVD):

    noise          = sample_latent_noise(actions, ...)                     train_svd.py:873   (pipeline:750-786)
    cond_sigmas    = rand_log_normal(loc=-3.0, scale=0.5)                  :876-881           (noise on the cond image)
    sigmas         = rand_log_normal(loc=0.7, scale=1.6)                   :886
    noisy          = latents + noise * sigmas                              :890
    timesteps      = 0.25 * log(sigmas)                                    :891-892
    inp            = cat([noisy / sqrt(sigmas^2 + 1), cond_latents], C)    :894, 926-931
    (ehs, cond_latents, action_ids) = apply_conditioning_dropout(...)      :911-921           (svd_utils.py:176-241)
    model_pred     = unet(inp, timesteps, ehs, added_time_ids, action_ids) :935-939
    denoised       = model_pred * c_out + c_skip * noisy                   :942-944   c_out = -s/sqrt(s^2+1), c_skip = 1/(s^2+1)
    loss           = mean(weighing * (denoised - latents)^2),  weighing = (1 + s^2) / s^2      :945-952
    loss.backward(); optimizer.step()  (torch.optim.AdamW, :653, 1123-1130)                    :961-969

Pinned by `tests/golden/train_step_tiny.npz`, produced by running the REFERENCE UNet class under torch autograd with the
same inputs (`oracle/make_train_golden.py`): loss and parameter gradients agree to fp32 round-off
(`tests/test_train_oracle.py`).  The reference supports batch size 1 per GPU in this loop ("TODO: support batch > 1",
:877); so does this restatement.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

import svd_oracle as O


def rand_log_normal(shape, loc: float = 0.0, scale: float = 1.0, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """train_svd.py:69-72 (k-diffusion): exp(Normal(loc, scale).icdf(u)), u uniform in [1e-7, 1 - 1e-7]."""
    u = torch.rand(shape, generator=generator) * (1 - 2e-7) + 1e-7
    return torch.distributions.Normal(loc, scale).icdf(u).exp()


def apply_conditioning_dropout(ehs: torch.Tensor, cond_latents: torch.Tensor, action_ids: torch.Tensor,
                               prob: Optional[float], random_p: torch.Tensor):
    """utils/svd_utils.py:176-241 with the uniform draw `random_p` (bsz,) made explicit: the image embedding is zeroed
    where p < 2*prob, the conditioning latents where prob <= p < 3*prob; the action branch is a no-op there (:238-239)."""
    if prob is None:
        return ehs, cond_latents, action_ids
    bsz = ehs.shape[0]
    prompt_mask = (random_p < 2 * prob).reshape(bsz, 1, 1)
    ehs = torch.where(prompt_mask, torch.zeros_like(ehs), ehs)
    image_mask = 1.0 - ((random_p >= prob).to(cond_latents.dtype) * (random_p < 3 * prob).to(cond_latents.dtype))
    cond_latents = image_mask.reshape(bsz, 1, 1, 1) * cond_latents
    return ehs, cond_latents, action_ids


def edm_training_inputs(latents: torch.Tensor, noise: torch.Tensor, sigmas: torch.Tensor,
                        cond_latents: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """latents, noise (bsz,T,4,h,w); sigmas (bsz,); cond_latents (bsz,4,h,w) -> (UNet input (bsz,T,8,h,w), timesteps
    (bsz,), noisy latents).  train_svd.py:888-894, 926-931."""
    s = sigmas.reshape(-1, 1, 1, 1, 1)
    noisy = latents + noise * s
    timesteps = 0.25 * sigmas.log()
    inp = noisy / (s ** 2 + 1) ** 0.5
    cond = cond_latents.unsqueeze(1).repeat(1, latents.shape[1], 1, 1, 1)
    return torch.cat([inp, cond], dim=2), timesteps, noisy


def edm_loss(model_pred: torch.Tensor, noisy: torch.Tensor, latents: torch.Tensor, sigmas: torch.Tensor) -> torch.Tensor:
    """train_svd.py:940-952."""
    s = sigmas.reshape(-1, 1, 1, 1, 1)
    c_out = -s / (s ** 2 + 1) ** 0.5
    c_skip = 1 / (s ** 2 + 1)
    denoised = model_pred * c_out + c_skip * noisy
    weighing = (1 + s ** 2) * s ** -2.0
    loss = torch.mean((weighing.float() * (denoised.float() - latents.float()) ** 2).reshape(latents.shape[0], -1), dim=1)
    return loss.mean()


def training_step(sd: Dict[str, torch.Tensor], cfg: dict, latents: torch.Tensor, noise: torch.Tensor, sigmas: torch.Tensor,
                  cond_latents: torch.Tensor, ehs: torch.Tensor, noise_aug_strength: float, action_ids: torch.Tensor,
                  dropout_prob: Optional[float] = None, random_p: Optional[torch.Tensor] = None):
    """One forward + backward of the loop body (batch size 1, as the reference).  `sd`: fp32 state dict; every tensor gets
    a gradient.  Returns (loss, model_pred, {name: grad})."""
    assert latents.shape[0] == 1, "the reference loop supports one sample per GPU (train_svd.py:877)"
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    ehs, cond_latents, action_ids = apply_conditioning_dropout(ehs, cond_latents, action_ids, dropout_prob, random_p)
    inp, timesteps, noisy = edm_training_inputs(latents, noise, sigmas, cond_latents)
    tids = torch.tensor([[7.0, 127.0, float(noise_aug_strength)]]).repeat(latents.shape[0], 1)   # :899-906
    pred = O.unet_forward(params, cfg, inp, timesteps.reshape(()), ehs, tids, action_ids)
    loss = edm_loss(pred, noisy, latents, sigmas)
    grads = torch.autograd.grad(loss, list(params.values()), allow_unused=True)
    return loss.detach(), pred.detach(), {k: g for k, g in zip(params, grads)}


def adamw_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr: float, beta1: float = 0.9,
               beta2: float = 0.999, eps: float = 1e-8, weight_decay: float = 1e-2):
    """torch.optim.AdamW update of one tensor (decoupled weight decay; train_svd.py:653, 1123-1130 defaults :283-300)."""
    p = p * (1 - lr * weight_decay)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    mh, vh = m / (1 - beta1 ** step), v / (1 - beta2 ** step)
    return p - lr * mh / (vh.sqrt() + eps), m, v
