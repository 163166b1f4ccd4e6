#!/usr/bin/env python
"""Generate tests/golden/train_step_tiny.npz by running the REFERENCE UNet class under torch autograd (build container only).

TEST INFRASTRUCTURE.  `FTsvd/train_svd.py` itself is not importable here (cv2, dataset modules), and its step is the body
of a loop, not a function: this script executes that body's formulas (train_svd.py:886-952, quoted line by line below)
around the reference's own `UNetSpatioTemporalConditionModel` (imported from /root/reference, seeded weights), its own
`get_action_ids` and `apply_conditioning_dropout` (utils/svd_utils.py), calls `loss.backward()`, and stores the inputs, the
loss, the model prediction and the parameter gradients: every gradient's norm, and the full gradient of a dozen tensors that
cover every operator class on the path.  `oracle/train_oracle.py` must reproduce them (tests/test_train_oracle.py).

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_train_golden.py          # scenario a: tests/golden/train_step_tiny.npz
    PYTHONDONTWRITEBYTECODE=1 python oracle/make_train_golden.py extra    # scenarios b, c: train_step_tiny_bc.npz
    PYTHONDONTWRITEBYTECODE=1 python oracle/make_train_golden.py full     # SERVED width: train_step_full_32x64.npz
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")

import numpy as np  # noqa: E402
import torch  # noqa: E402

from _ref_import import import_reference  # noqa: E402
import wiw_amd  # noqa: E402,F401
from make_golden import ref_unet, save  # noqa: E402
from wiw_amd.config import UNetConfig  # noqa: E402

# full gradients are stored for these tensors (one per operator class); norms for all
FULL = ["conv_in.weight", "conv_out.bias", "time_embedding.linear_1.weight", "add_action_proj.proj.weight",
        "down_blocks.0.resnets.0.spatial_res_block.conv1.weight", "down_blocks.0.resnets.0.spatial_res_block.norm2.weight",
        "down_blocks.0.resnets.0.temporal_res_block.conv2.weight", "down_blocks.0.resnets.0.time_mixer.mix_factor",
        "down_blocks.1.attentions.0.transformer_blocks.0.attn1.to_q.weight",
        "down_blocks.1.attentions.0.transformer_blocks.0.norm3.bias",
        "down_blocks.1.attentions.0.transformer_blocks.0.ff.net.0.proj.weight",
        "down_blocks.1.attentions.0.temporal_transformer_blocks.0.attn1.to_v.weight",
        "down_blocks.1.attentions.0.temporal_transformer_blocks.0.attn2.to_out.0.bias",
        "down_blocks.1.attentions.0.time_pos_embed.linear_2.weight",
        "up_blocks.1.resnets.0.spatial_res_block.conv_shortcut.weight", "up_blocks.0.upsamplers.0.conv.weight",
        "down_blocks.0.downsamplers.0.conv.weight", "mid_block.attentions.0.proj_out.weight"]


def main(full_width: bool = False):
    """full_width: the SERVED architecture (320 / 640 / 1280 / 1280, 5 / 10 / 20 / 20 heads, T = 14) at a 32 x 64 latent — the
    same step, every gradient norm, and of the FULL tensors either the whole gradient (<= 16 384 elements) or its first 8
    rows (the fixture stays ~1 MB; the un-stored rows are covered by the tensor's norm)."""
    ns = import_reference()
    from utils.svd_utils import apply_conditioning_dropout  # type: ignore  (reference)

    torch.set_num_threads(8)
    cfg = UNetConfig() if full_width else UNetConfig.tiny(4)
    T, h, w = cfg.num_frames, 32, 64   # L3 is 4 x 8 = 32 sites: the attention backward kernels need S % 16 == 0
    m = ref_unet(ns, cfg, seed=7).float().train()
    for prm in m.parameters():
        prm.requires_grad_(True)
    rs = np.random.RandomState(17)
    latents = torch.from_numpy(rs.standard_normal((1, T, 4, h, w)).astype(np.float32)) * 0.8
    noise = torch.from_numpy(rs.standard_normal((1, T, 4, h, w)).astype(np.float32))
    conditional_latents = torch.from_numpy(rs.standard_normal((1, 4, h, w)).astype(np.float32))
    encoder_hidden_states = torch.from_numpy(rs.standard_normal((1, 1, cfg.cross_attention_dim)).astype(np.float32))
    actions = np.array([[4, 2, 1, 3] + ([1, 1, 3, 2, 1, 2, 3, 1, 1, 2] if full_width else [])], dtype=np.int64)[:, :T]
    sigmas = torch.tensor([1.7], dtype=torch.float32)           # one draw of rand_log_normal(loc=0.7, scale=1.6), fixed
    noise_aug_strength = 0.043                                   # one draw of rand_log_normal(loc=-3.0, scale=0.5), fixed
    random_p = torch.tensor([0.9])                               # conditioning dropout draw: keeps every condition
    bsz = 1

    # ---- train_svd.py:888-894
    sig = sigmas[:, None, None, None, None]
    noisy_latents = latents + noise * sig
    timesteps = torch.Tensor([0.25 * sigma.log() for sigma in sig])
    inp_noisy_latents = noisy_latents / ((sig ** 2 + 1) ** 0.5)
    # ---- :899-907
    added_time_ids = torch.tensor([[7, 127, noise_aug_strength]], dtype=torch.float32).repeat(bsz, 1)
    action_ids = ns.get_action_ids(bsz, torch.from_numpy(actions), "micro_cond", torch.float32)
    # ---- :911-921 (the reference function; its uniform draw replaced by the fixed `random_p` through a seeded generator stub)
    class _Gen:  # torch.rand(bsz, device=..., generator=_) -> random_p
        pass
    orig_rand = torch.rand
    torch.rand = lambda *a, **k: random_p.clone()
    try:
        ehs_d, cond_d, act_d = apply_conditioning_dropout(
            encoder_hidden_states=encoder_hidden_states, conditional_latents=conditional_latents,
            action_conditioning=action_ids, bsz=bsz, conditioning_dropout_prob=0.1, generator=None)
    finally:
        torch.rand = orig_rand
    # ---- :926-931
    cond_rep = cond_d.unsqueeze(1).repeat(1, noisy_latents.shape[1], 1, 1, 1)
    inp = torch.cat([inp_noisy_latents, cond_rep], dim=2)
    # ---- :933-939
    target = latents
    model_pred = m(inp, timesteps, ehs_d, added_time_ids=added_time_ids, added_action_ids=act_d).sample
    # ---- :941-952
    c_out = -sig / ((sig ** 2 + 1) ** 0.5)
    c_skip = 1 / (sig ** 2 + 1)
    denoised_latents = model_pred * c_out + c_skip * noisy_latents
    weighing = (1 + sig ** 2) * (sig ** -2.0)
    loss = torch.mean((weighing.float() * (denoised_latents.float() - target.float()) ** 2).reshape(target.shape[0], -1), dim=1)
    loss = loss.mean()
    loss.backward()                                               # :962
    names, norms, full = [], [], {}
    for k, prm in m.named_parameters():
        g = prm.grad
        names.append(k)
        norms.append(0.0 if g is None else float(g.double().norm()))
        if k in FULL:
            if full_width and g.numel() > 16384:
                full["grad8__" + k.replace(".", "__")] = g.reshape(g.shape[0], -1)[:8].numpy().copy()
            else:
                full["grad__" + k.replace(".", "__")] = g.numpy()
    missing = [k for k in FULL if not any(p + k.replace(".", "__") in full for p in ("grad__", "grad8__"))]
    assert not missing, missing
    # one AdamW step of the reference optimiser class on two tensors (torch.optim.AdamW, train_svd.py:653, 1123-1130)
    opt_names = ["conv_in.weight", "down_blocks.1.attentions.0.transformer_blocks.0.norm3.bias"]
    prms = dict(m.named_parameters())
    opt = torch.optim.AdamW([prms[k] for k in opt_names], lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    before = {k: prms[k].detach().clone() for k in opt_names}
    opt.step()
    save("train_step_full_32x64.npz" if full_width else "train_step_tiny.npz", weight_seed=np.array(7), latents=latents.numpy(), noise=noise.numpy(),
         conditional_latents=conditional_latents.numpy(), encoder_hidden_states=encoder_hidden_states.numpy(),
         actions=actions, action_ids=action_ids.numpy(), sigmas=sigmas.numpy(), noise_aug_strength=np.array(noise_aug_strength),
         random_p=random_p.numpy(), dropout_prob=np.array(0.1), loss=np.array(float(loss)), model_pred=model_pred.detach().numpy(),
         grad_names=np.array(names), grad_norms=np.array(norms),
         adamw_names=np.array(opt_names), adamw_lr=np.array(1e-3),
         **{"adamw_before__" + k.replace(".", "__"): before[k].numpy() for k in opt_names},
         **{"adamw_after__" + k.replace(".", "__"): prms[k].detach().numpy() for k in opt_names}, **full)


def derived_inputs(g, which: str):
    """Inputs of the extra scenarios, derived from the stored scenario-a tensors so that no second set of inputs has to be
    committed (tests rebuild them the same way): flips and rescalings."""
    lat, noise = torch.from_numpy(g["latents"]), torch.from_numpy(g["noise"])
    cond, ehs = torch.from_numpy(g["conditional_latents"]), torch.from_numpy(g["encoder_hidden_states"])
    if which == "b":
        return lat.flip(-1) * 0.9, noise.flip(-2), cond * 1.1, ehs.flip(-1)
    return lat.flip(-2) * 1.1, noise.flip(-1), cond.flip(-1) * 0.8, ehs * 0.7


SCENARIOS = {   # sigma, noise_aug_strength, random_p (dropout prob 0.1), actions
    "b": (0.35, 0.021, 0.25, [[1, 3, 3, 2]]),   # low noise level; prob <= p < 3 prob and p >= 2 prob: conditioning LATENTS zeroed
    "c": (6.5, 0.090, 0.05, [[2, 2, 4, 1]]),    # high noise level; p < 2 prob and p < prob: image EMBEDDING zeroed, latents kept
}
FULL_EXTRA = ["conv_in.weight", "down_blocks.1.attentions.0.transformer_blocks.0.attn1.to_q.weight",
              "down_blocks.0.resnets.0.temporal_res_block.conv2.weight"]


def extra():
    """tests/golden/train_step_tiny_bc.npz: two more seeded steps of the reference (other noise levels, the two
    conditioning-dropout branches that zero a condition, other actions): loss, prediction, every gradient norm, three
    full gradients.  Same weights (seed 7) and the scenario-a inputs transformed by `derived_inputs`."""
    ns = import_reference()
    from utils.svd_utils import apply_conditioning_dropout  # type: ignore  (reference)

    torch.set_num_threads(8)
    cfg = UNetConfig.tiny(4)
    g = np.load(os.path.join(ROOT, "tests", "golden", "train_step_tiny.npz"))
    out = {}
    for which, (sigma, nas, p, actions) in SCENARIOS.items():
        m = ref_unet(ns, cfg, seed=7).float().train()
        for prm in m.parameters():
            prm.requires_grad_(True)
        latents, noise, conditional_latents, encoder_hidden_states = derived_inputs(g, which)
        sigmas, random_p, bsz = torch.tensor([sigma], dtype=torch.float32), torch.tensor([p]), 1
        sig = sigmas[:, None, None, None, None]
        noisy_latents = latents + noise * sig                                                     # train_svd.py:888-894
        timesteps = torch.Tensor([0.25 * s_.log() for s_ in sig])
        inp_noisy_latents = noisy_latents / ((sig ** 2 + 1) ** 0.5)
        added_time_ids = torch.tensor([[7, 127, nas]], dtype=torch.float32).repeat(bsz, 1)          # :899-907
        action_ids = ns.get_action_ids(bsz, torch.from_numpy(np.array(actions, dtype=np.int64)), "micro_cond", torch.float32)
        orig_rand = torch.rand
        torch.rand = lambda *a, **k: random_p.clone()
        try:                                                                                       # :911-921
            ehs_d, cond_d, act_d = apply_conditioning_dropout(
                encoder_hidden_states=encoder_hidden_states, conditional_latents=conditional_latents,
                action_conditioning=action_ids, bsz=bsz, conditioning_dropout_prob=0.1, generator=None)
        finally:
            torch.rand = orig_rand
        assert (which == "b") == bool((cond_d == 0).all()) and (which == "c") == bool((ehs_d == 0).all())
        cond_rep = cond_d.unsqueeze(1).repeat(1, noisy_latents.shape[1], 1, 1, 1)                 # :926-931
        inp = torch.cat([inp_noisy_latents, cond_rep], dim=2)
        model_pred = m(inp, timesteps, ehs_d, added_time_ids=added_time_ids, added_action_ids=act_d).sample   # :933-939
        c_out = -sig / ((sig ** 2 + 1) ** 0.5)                                                    # :941-952
        c_skip = 1 / (sig ** 2 + 1)
        denoised_latents = model_pred * c_out + c_skip * noisy_latents
        weighing = (1 + sig ** 2) * (sig ** -2.0)
        loss = torch.mean((weighing.float() * (denoised_latents.float() - latents.float()) ** 2).reshape(bsz, -1), dim=1).mean()
        loss.backward()
        names, norms = [], []
        for k, prm in m.named_parameters():
            names.append(k)
            norms.append(0.0 if prm.grad is None else float(prm.grad.double().norm()))
            if k in FULL_EXTRA:
                out[f"{which}__grad__" + k.replace(".", "__")] = prm.grad.numpy()
        out.update({f"{which}__sigmas": sigmas.numpy(), f"{which}__noise_aug_strength": np.array(nas), f"{which}__random_p": random_p.numpy(),
                    f"{which}__action_ids": action_ids.numpy(), f"{which}__loss": np.array(float(loss)),
                    f"{which}__model_pred": model_pred.detach().numpy(), f"{which}__grad_norms": np.array(norms)})
        out["grad_names"] = np.array(names)
        print(which, "loss", float(loss))
    save("train_step_tiny_bc.npz", dropout_prob=np.array(0.1), weight_seed=np.array(7), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "extra":
        extra()
    elif len(sys.argv) > 1 and sys.argv[1] == "full":
        main(full_width=True)
    else:
        main()
