"""First kernels of row f2 (the fine-tuning step): AdamW and the EDM loss + gradient, against torch / the reference-pinned
fixture `tests/golden/train_step_tiny.npz`.  The operators' backward kernels do not exist yet."""
import numpy as np
import pytest
import torch

import train_oracle as TO

DEV = "cuda:0"


def test_prepare_step_matches_the_oracle(golden):
    """Host-side step preparation (CPU): product `train.prepare_step` == oracle `edm_training_inputs` on the fixture."""
    import wiw_amd  # noqa: F401
    from wiw_amd import train as T

    g = golden("train_step_tiny.npz")
    lat, noise, cl = (torch.from_numpy(g[k]) for k in ("latents", "noise", "conditional_latents"))
    ehs, aid = torch.from_numpy(g["encoder_hidden_states"]), torch.from_numpy(g["action_ids"])
    st = T.prepare_step(lat, noise, float(g["sigmas"][0]), cl, ehs, float(g["noise_aug_strength"]), aid,
                        dropout_prob=float(g["dropout_prob"]), random_p=torch.from_numpy(g["random_p"]))
    inp, t, noisy = TO.edm_training_inputs(lat, noise, torch.from_numpy(g["sigmas"]), cl)
    assert torch.allclose(st.unet_input, inp, rtol=0, atol=1e-6) and abs(st.timestep - float(t)) < 1e-6
    assert torch.equal(st.noisy, noisy) and st.added_time_ids.tolist()[0][:2] == [7.0, 127.0]
    # dropout branches
    e, c, _ = T.apply_conditioning_dropout(torch.ones(2, 1, 4), torch.ones(2, 4, 2, 2), aid, 0.1, torch.tensor([0.05, 0.25]))
    assert e[:, 0, 0].tolist() == [0.0, 1.0] and c[:, 0, 0, 0].tolist() == [1.0, 0.0]
    gen = torch.Generator().manual_seed(3)
    s = T.rand_log_normal((10000,), 0.7, 1.6, generator=gen)
    assert abs(float(s.log().mean()) - 0.7) < 0.06 and abs(float(s.log().std()) - 1.6) < 0.06


@pytest.mark.gpu
def test_edm_loss_and_grad_kernel(golden):
    import wiw_amd  # noqa: F401
    from wiw_amd import train as T
    from wiw_amd.hip import Hip

    g = golden("train_step_tiny.npz")
    hip = Hip(torch.device(DEV))
    lat, noise, cl = (torch.from_numpy(g[k]) for k in ("latents", "noise", "conditional_latents"))
    st = T.prepare_step(lat, noise, float(g["sigmas"][0]), cl, torch.from_numpy(g["encoder_hidden_states"]),
                        float(g["noise_aug_strength"]), torch.from_numpy(g["action_ids"]))
    pred = torch.from_numpy(g["model_pred"])
    loss, grad = T.TrainStep(hip).loss_and_grad(pred, st)
    assert abs(float(loss) - float(g["loss"])) <= 2e-6 * float(g["loss"])       # the reference's loss on its own prediction
    p = pred.clone().requires_grad_(True)
    ref = TO.edm_loss(p, st.noisy, lat, torch.from_numpy(g["sigmas"]))
    ref.backward()
    assert float((grad.cpu() - p.grad).abs().max()) <= 1e-6 * float(p.grad.abs().max())
    with pytest.raises(NotImplementedError):
        T.TrainStep(hip).backward()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_adamw_kernel(golden, dtype):
    import wiw_amd  # noqa: F401
    from wiw_amd.hip import Hip

    g = golden("train_step_tiny.npz")
    hip = Hip(torch.device(DEV), dtype)
    # (a) the reference optimiser's first step on two tensors of the fixture
    for name in [str(n) for n in g["adamw_names"]]:
        key = name.replace(".", "__")
        p = torch.from_numpy(g["adamw_before__" + key]).reshape(-1).to(DEV).clone()
        grad = torch.from_numpy(g["grad__" + key]).reshape(-1).to(DEV)
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        p16 = torch.empty(p.numel(), dtype=dtype, device=DEV)
        hip.adamw_step(p, grad, m, v, 1, float(g["adamw_lr"]), p16=p16)
        after = torch.from_numpy(g["adamw_after__" + key]).reshape(-1)
        assert float((p.cpu() - after).abs().max()) <= 2e-6 * float(after.abs().max()) + 1e-9, name
        assert torch.equal(p16.cpu(), p.cpu().to(dtype))
    # (b) five steps against torch.optim.AdamW on an odd-sized range (tail path), bias correction included
    n = 100003
    gen = torch.Generator().manual_seed(0)
    p0 = torch.randn(n, generator=gen)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([ref], lr=3e-3, betas=(0.9, 0.95), weight_decay=0.05, eps=1e-8)
    p, m, v = p0.to(DEV).clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 6):
        gr = torch.randn(n, generator=gen)
        ref.grad = gr.clone()
        opt.step()
        hip.adamw_step(p, gr.to(DEV), m, v, step, 3e-3, beta1=0.9, beta2=0.95, eps=1e-8, weight_decay=0.05)
    assert float((p.cpu() - ref.detach()).abs().max()) <= 5e-6
