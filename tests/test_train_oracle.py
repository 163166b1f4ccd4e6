"""Row f2 (the fine-tuning step, FTsvd/train_svd.py:844-970) — oracle side only, CPU.

`oracle/train_oracle.py` (EDM pre-conditioning, conditioning dropout, loss, autograd through the oracle UNet, AdamW) against
`tests/golden/train_step_tiny.npz`, which `oracle/make_train_golden.py` produced by running the REFERENCE UNet class, its
`get_action_ids` / `apply_conditioning_dropout` and `torch.optim.AdamW` under autograd.  This pins the checker of the
backward kernels before any of them exists (f2 is not built yet).

Tolerances: fp32 summation order only — loss 1e-5 relative, prediction and gradients 2e-4 of the tensor's max."""
import numpy as np
import pytest
import torch

import svd_oracle as O
import train_oracle as TO


def _setup(golden):
    import wiw_amd  # noqa: F401
    from wiw_amd.config import UNetConfig
    from wiw_amd.weights import random_state_dict

    g = golden("train_step_tiny.npz")
    cfg = UNetConfig.tiny(4)
    sd = {k: torch.from_numpy(v) for k, v in random_state_dict(cfg, int(g["weight_seed"])).items()}
    return g, cfg, sd


def test_training_step_matches_reference_autograd(golden):
    g, cfg, sd = _setup(golden)
    torch.set_num_threads(8)
    aid = torch.from_numpy(O.action_ids_idx_encode(g["actions"])).float()
    assert np.array_equal(aid.numpy(), g["action_ids"])
    loss, pred, grads = TO.training_step(
        sd, cfg.as_dict(), torch.from_numpy(g["latents"]), torch.from_numpy(g["noise"]), torch.from_numpy(g["sigmas"]),
        torch.from_numpy(g["conditional_latents"]), torch.from_numpy(g["encoder_hidden_states"]),
        float(g["noise_aug_strength"]), aid, dropout_prob=float(g["dropout_prob"]), random_p=torch.from_numpy(g["random_p"]))
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"])), (float(loss), float(g["loss"]))
    ref = torch.from_numpy(g["model_pred"])
    assert float((pred - ref).abs().max()) <= 2e-4 * float(ref.abs().max())
    # every parameter's gradient norm (a parameter the loss does not reach has norm 0 on both sides)
    names, norms = [str(n) for n in g["grad_names"]], g["grad_norms"]
    assert set(names) <= set(grads), sorted(set(names) - set(grads))[:5]
    # The single-key cross-attention makes norm2 and attn2.to_q / to_k DEAD in the backward pass too: the
    # softmax over one key is identically 1, its gradient identically 0.  The reference's autograd leaves 1e-9-level
    # round-off there (nine orders below the live gradients); the oracle, which never evaluates them, returns none.
    # ... and `add_embedding` (fps / motion bucket) is overwritten for micro_cond (unet:482): no gradient either
    is_dead = lambda n: ("transformer_blocks.0.norm2." in n) or (".attn2.to_q." in n) or (".attn2.to_k." in n) or \
        n.startswith("add_embedding.")  # noqa: E731
    med = float(np.median(norms[norms > 0]))
    worst, dead = 0.0, 0
    for n, nr in zip(names, norms):
        gn = 0.0 if grads[n] is None else float(grads[n].double().norm())
        if is_dead(n):
            assert gn == 0.0 and nr <= 1e-3 * med, (n, gn, nr)
            dead += 1
        else:
            assert nr > 0, n
            # (a scalar mix_factor whose gradient is a cancelling sum of 1e-8 gets an absolute allowance)
            worst = max(worst, max(abs(gn - nr) - 1e-9, 0.0) / nr)
    print(f"[f2 oracle] {len(names)} gradient norms ({dead} dead cross-attention / add_embedding parameters), worst relative "
          f"deviation of the live ones {worst:.2e}")
    assert dead == 132 and worst <= 1e-3
    # full gradients of one tensor per operator class
    n_full = 0
    for key in g.files:
        if key.startswith("grad__"):
            name = key[len("grad__"):].replace("__", ".")
            r = torch.from_numpy(g[key])
            assert float((grads[name] - r).abs().max()) <= 2e-4 * float(r.abs().max()) + 1e-12, name
            n_full += 1
    assert n_full >= 15


def test_conditioning_dropout_cases():
    ehs, cl, act = torch.ones(4, 1, 8), torch.ones(4, 4, 2, 2), torch.ones(4, 3, 3)
    p = torch.tensor([0.05, 0.15, 0.25, 0.9])          # prob 0.1: image-embedding only | both | latents only | none
    e, c, a = TO.apply_conditioning_dropout(ehs, cl, act, 0.1, p)
    assert e[:, 0, 0].tolist() == [0.0, 0.0, 1.0, 1.0] and c[:, 0, 0, 0].tolist() == [1.0, 0.0, 0.0, 1.0]
    assert torch.equal(a, act)
    assert TO.apply_conditioning_dropout(ehs, cl, act, None, p)[0] is ehs


def test_edm_inputs_and_loss_identities():
    """c_skip * noisy + c_out * v with the exact v-prediction of the clean latents gives zero loss; sigma -> timestep."""
    g = torch.Generator().manual_seed(0)
    lat, noise = torch.randn(1, 4, 4, 8, 8, generator=g), torch.randn(1, 4, 4, 8, 8, generator=g)
    s = TO.rand_log_normal((1,), 0.7, 1.6, generator=g)
    inp, t, noisy = TO.edm_training_inputs(lat, noise, s, torch.randn(1, 4, 8, 8, generator=g))
    assert inp.shape == (1, 4, 8, 8, 8) and abs(float(t) - 0.25 * float(s.log())) < 1e-6
    v_exact = (noisy / (s ** 2 + 1) - lat) * ((s ** 2 + 1) ** 0.5 / s)       # solves denoised == latents
    assert float(TO.edm_loss(v_exact, noisy, lat, s)) < 1e-10
    assert float(TO.edm_loss(torch.zeros_like(lat), noisy, lat, s)) > 0


def test_adamw_matches_torch_optim(golden):
    g, cfg, sd = _setup(golden)
    for name in [str(n) for n in g["adamw_names"]]:
        key = name.replace(".", "__")
        before, after = torch.from_numpy(g["adamw_before__" + key]), torch.from_numpy(g["adamw_after__" + key])
        grad = torch.from_numpy(g["grad__" + key])
        p, m, v = TO.adamw_step(before, grad, torch.zeros_like(before), torch.zeros_like(before), 1, float(g["adamw_lr"]))
        assert float((p - after).abs().max()) <= 1e-6 * float(after.abs().max()) + 1e-9, name
        assert torch.equal(before, sd[name])


def test_lr_schedules_match_the_published_lambda_schedulers():
    """`train.lr_at` against `transformers.get_scheduler` — the same LambdaLR multipliers `diffusers.optimization.get_scheduler`
    applies in train_svd.py:1131-1136 (diffusers is not in this image; its schedule functions are the transformers ones)."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import wiw_amd  # noqa: F401
    from transformers import get_scheduler
    from wiw_amd.train import lr_at

    for name in ("constant", "constant_with_warmup", "linear", "cosine"):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.AdamW([p], lr=2e-5)
        sch = get_scheduler(name, optimizer=opt, num_warmup_steps=50, num_training_steps=120)
        for step in range(125):
            assert abs(opt.param_groups[0]["lr"] - lr_at(step, 2e-5, name, 50, 120)) <= 1e-12, (name, step)
            opt.step()
            sch.step()


def _derived(g, which):
    """The inputs of scenarios b / c: the stored scenario-a tensors transformed as oracle/make_train_golden.py `derived_inputs`."""
    lat, noise = torch.from_numpy(g["latents"]), torch.from_numpy(g["noise"])
    cond, ehs = torch.from_numpy(g["conditional_latents"]), torch.from_numpy(g["encoder_hidden_states"])
    if which == "b":
        return lat.flip(-1) * 0.9, noise.flip(-2), cond * 1.1, ehs.flip(-1)
    return lat.flip(-2) * 1.1, noise.flip(-1), cond.flip(-1) * 0.8, ehs * 0.7


@pytest.mark.parametrize("which", ["b", "c"])
def test_training_step_other_noise_levels_and_dropout_branches(golden, which):
    """Two more steps of the REFERENCE (tests/golden/train_step_tiny_bc.npz): sigma 0.35 with the conditioning latents
    dropped (b), sigma 6.5 with the image embedding dropped (c), other actions — loss, prediction, every gradient norm and
    three full gradients of the restatement against the reference's autograd."""
    g, cfg, sd = _setup(golden)
    e = golden("train_step_tiny_bc.npz")
    torch.set_num_threads(8)
    lat, noise, cond, ehs = _derived(g, which)
    loss, pred, grads = TO.training_step(sd, cfg.as_dict(), lat, noise, torch.from_numpy(e[f"{which}__sigmas"]), cond, ehs,
                                         float(e[f"{which}__noise_aug_strength"]), torch.from_numpy(e[f"{which}__action_ids"]),
                                         dropout_prob=float(e["dropout_prob"]), random_p=torch.from_numpy(e[f"{which}__random_p"]))
    assert abs(float(loss) - float(e[f"{which}__loss"])) <= 1e-5 * float(e[f"{which}__loss"])
    ref = torch.from_numpy(e[f"{which}__model_pred"])
    assert float((pred - ref).abs().max()) <= 2e-4 * float(ref.abs().max())
    is_dead = lambda n: ("transformer_blocks.0.norm2." in n) or (".attn2.to_q." in n) or (".attn2.to_k." in n) or \
        n.startswith("add_embedding.")  # noqa: E731
    worst = 0.0
    for n, nr in zip((str(n) for n in e["grad_names"]), e[f"{which}__grad_norms"]):
        if is_dead(n):
            continue
        gn = 0.0 if grads[n] is None else float(grads[n].double().norm())
        if which == "c" and (".attn2." in n or "norm2" in n) and nr == 0.0:
            assert gn == 0.0, n                        # zeroed image embedding: the cross-attention value path gets no gradient
            continue
        worst = max(worst, max(abs(gn - nr) - 1e-9, 0.0) / max(nr, 1e-12))
    assert worst <= 1e-3, worst
    for key in e.files:
        if key.startswith(f"{which}__grad__"):
            name = key[len(f"{which}__grad__"):].replace("__", ".")
            r = torch.from_numpy(e[key])
            assert float((grads[name] - r).abs().max()) <= 2e-4 * float(r.abs().max()) + 1e-12, name


def test_weight_gradient_plan_functions_return_launchable_plans():
    """`train.wgrad_plan` / `wgrad_tn_splits` / `wgrad_default_plan` (pure host logic) over the served shapes and odd ones: a
    split-K factor always divides the K-tile count and leaves >= 8 tiles per range (what wiw_gemm_bf16 requires), a row split
    keeps >= 128 rows (or is 1), linear layers default to the row-major kernel, convolutions to the transposed-operand GEMM."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import wiw_amd  # noqa: F401
    from wiw_amd.train import wgrad_default_plan, wgrad_plan, wgrad_tn_splits

    M0 = 14 * 72 * 128
    for n_out in (64, 320, 640, 960, 1280, 1920, 2560, 5120, 10240):
        for k_in in (192, 320, 576, 640, 1280, 2880, 5760, 11520, 23040):
            for rows in (64, 2048, M0 >> 4, M0 >> 2, M0):
                flip, sk = wgrad_plan(n_out, k_in, rows)
                nk = rows // 64
                assert flip in (0, 1) and sk >= 1 and nk % sk == 0 and (sk == 1 or nk // sk >= 8), (n_out, k_in, rows, flip, sk)
                sp = wgrad_tn_splits(n_out, k_in, rows)
                assert sp == 1 or rows // sp >= 128, (n_out, k_in, rows, sp)
                mode, s = wgrad_default_plan(n_out, k_in, rows, conv=False)
                assert mode in (2, 3) and s >= 1
                assert wgrad_default_plan(n_out, k_in, rows, conv=True) == wgrad_plan(n_out, k_in, rows)
    # a 320-wide dy against a wide x: operands swapped (3 x 128 columns instead of 2 x 256 rows); square / tall: not
    assert wgrad_default_plan(320, 1280, M0, False)[0] == 3 and wgrad_default_plan(320, 320, M0, False)[0] == 2
    assert wgrad_default_plan(2560, 320, M0, False)[0] == 2


def test_batch_from_pixels_composition_on_a_cpu_front():
    """`train.batch_from_pixels` (train_svd.py:846-931) with a CPU stand-in for the HIP front end (the oracle's VAE encoder, a
    fixed embedding): the host logic — which draws are made, how the latents, the conditioning sample, the rotated noise, the
    EDM input scaling, the time ids and the action ids are put together — against the formulas of the reference's loop.  The HIP
    front end itself is pinned on the GPU (tests/test_hip_vae.py)."""
    import vae_oracle as VO
    import wiw_amd  # noqa: F401
    from wiw_amd import frontend as FE
    from wiw_amd import train as T
    from wiw_amd.pipeline import action_ids_idx_encode, rotate_latent_noise

    cfg = dict(block_out_channels=(32, 32, 64, 64), layers_per_block=1)
    sd = {k: torch.from_numpy(v) for k, v in FE.vae_random_state_dict(3, **cfg).items()}

    class Vae:
        device, scaling_factor = torch.device("cpu"), 0.18215

        def encode_moments(self, x):
            return VO.vae_encode_moments(sd, x, len(cfg["block_out_channels"]), cfg["layers_per_block"])

    class Front:
        vae = Vae()

        def _embed(self, x, clip_images):
            assert x.shape == (1, 3, 64, 128) and clip_images is None
            return torch.full((1, 1, 1024), 0.25)

    g = torch.Generator().manual_seed(0)
    Tn, H, W = 4, 64, 128
    px = torch.tanh(torch.randn(1, Tn, 3, H, W, generator=g))
    past = torch.tanh(torch.randn(1, 1, 3, H, W, generator=g))
    acts = np.array([[1, 3, 1, 2]])
    h, w = H // 8, W // 8
    d = dict(vae_eps=torch.randn(Tn, 4, h, w, generator=g), cond_eps=torch.randn(1, 4, h, w, generator=g),
             pixel_noise=torch.randn(1, 3, H, W, generator=g), latent_noise=torch.randn(1, Tn, 4, h, w, generator=g), sigma=0.8,
             cond_sigma=0.05, random_p=torch.tensor([0.15]))
    st = T.batch_from_pixels(Front(), px, past, acts, dropout_prob=0.1, draws=d)
    m, lv = Front.vae.encode_moments(px[0])
    lat = ((m + torch.exp(0.5 * lv) * d["vae_eps"]) * 0.18215)[None]
    cm, cl = Front.vae.encode_moments(px[0, 0:1] + d["pixel_noise"] * 0.05)
    cond = cm + torch.exp(0.5 * cl) * d["cond_eps"]
    noise = rotate_latent_noise(d["latent_noise"], acts)
    assert torch.allclose(st.target, lat, atol=1e-6) and torch.allclose(st.noisy, lat + 0.8 * noise, atol=1e-6)
    # random_p = 0.15 with prob 0.1: in [prob, 3 prob) -> conditioning latents dropped, and < 2 prob -> image embedding zeroed
    want = torch.cat([(lat + 0.8 * noise) / (0.8 ** 2 + 1) ** 0.5, torch.zeros(1, Tn, 4, h, w)], dim=2)
    assert torch.allclose(st.unet_input, want, atol=1e-6) and float(st.ehs.abs().max()) == 0.0
    assert st.timestep == pytest.approx(0.25 * np.log(0.8)) and st.added_time_ids.tolist() == [[7.0, 127.0, pytest.approx(0.05)]]
    assert torch.equal(st.action_ids, torch.from_numpy(action_ids_idx_encode(acts)))
    keep = T.batch_from_pixels(Front(), px, past, acts, dropout_prob=0.1, draws=dict(d, random_p=torch.tensor([0.9])))
    assert torch.allclose(keep.unet_input[:, :, 4:], cond.unsqueeze(1).repeat(1, Tn, 1, 1, 1), atol=1e-6)
    assert float(keep.ehs[0, 0, 0]) == 0.25
    # every draw comes from the generator when none is injected: same seed, same batch
    a = T.batch_from_pixels(Front(), px, past, acts, dropout_prob=0.1, generator=torch.Generator().manual_seed(4))
    b = T.batch_from_pixels(Front(), px, past, acts, dropout_prob=0.1, generator=torch.Generator().manual_seed(4))
    assert torch.equal(a.unet_input, b.unet_input) and a.sigma == b.sigma and torch.equal(a.ehs, b.ehs)
