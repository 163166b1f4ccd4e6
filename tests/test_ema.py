"""`--use_ema` (train_svd.py:566-568, 979-980, 588-604): `train_unet.EMAShadow` against the REFERENCE's EMAModel, run here by
`oracle/make_golden.py ema` over a drifting tiny UNet with a frozen parameter (tests/golden/ema_steps.npz: the decay of every
step and the parameter / shadow trajectories, fp32 bit patterns), the unet_ema/ checkpoint directory, and — on the GPU — the
`wiw_ema_step_f32` kernel and the Trainer wiring."""
import json
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ema_steps.npz")


def _replay(ema, g, watch, params):
    for step in range(1, len(g["decays"]) + 1):
        for w in watch:
            params[w].copy_(torch.from_numpy(g[f"p{step}.{w}"]))
        ema.step(params)
        assert ema.cur_decay_value == float(g["decays"][step - 1]), (step, ema.cur_decay_value)
        for w in watch:
            assert np.array_equal(ema.shadow[w].cpu().numpy(), g[f"s{step}.{w}"]), (step, w)


def test_ema_shadow_reproduces_the_reference_bit_for_bit():
    import wiw_amd  # noqa: F401
    from wiw_amd.train_unet import EMAShadow

    g = np.load(GOLD)
    watch, frozen = [str(w) for w in g["watch"]], {str(f) for f in g["frozen"]}
    params = {w: torch.from_numpy(g[f"p0.{w}"]).clone() for w in watch}
    ema = EMAShadow(params, trainable=lambda n: n not in frozen)
    assert float(g["decays"][0]) == 0.0 and 0.6 < float(g["decays"][-1]) < 0.62      # (1 + s) / (10 + s), s = 13
    _replay(ema, g, watch, params)
    assert sorted(ema.state()) == [str(k) for k in g["state_keys"]]                 # what save_pretrained registers into the config
    # the frozen parameter's shadow IS the parameter (copied, not averaged); the others lag behind
    f = next(iter(frozen))
    assert torch.equal(ema.shadow[f], params[f])
    assert any(not torch.equal(ema.shadow[w], params[w]) for w in watch if w not in frozen)
    # store / copy_to / restore (validation under the averaged weights, train_svd.py:1004-1007, 1189-1191)
    before = {k: v.clone() for k, v in params.items()}
    ema.store(params); ema.copy_to(params)
    assert all(torch.equal(params[k], ema.shadow[k]) for k in params)
    ema.restore(params)
    assert all(torch.equal(params[k], before[k]) for k in params)


def test_ema_warmup_decay_schedule():
    """use_ema_warmup: 1 - (1 + s / inv_gamma) ** -power, capped by `decay`, floored by `min_decay` (training_utils.py:405-422)."""
    import wiw_amd  # noqa: F401
    from wiw_amd.train_unet import EMAShadow

    e = EMAShadow({"w": torch.zeros(4)}, use_ema_warmup=True, inv_gamma=1.0, power=2 / 3, decay=0.999, min_decay=0.1,
                  update_after_step=2)
    assert e.get_decay(1) == 0.0 and e.get_decay(3) == 0.0            # s = step - update_after_step - 1 <= 0
    assert e.get_decay(4) == pytest.approx(max(1 - 2 ** (-2 / 3), 0.1))
    assert e.get_decay(10 ** 9) == 0.999


def test_unet_ema_checkpoint_directory_round_trip(tmp_path):
    """checkpoint-<n>/unet_ema = the averaged weights under the reference's parameter names + a config.json that carries the
    architecture AND the EMA state (EMAModel.save_pretrained registers it into the config, training_utils.py:390-403)."""
    import wiw_amd  # noqa: F401
    from wiw_amd import checkpoint as C
    from wiw_amd.config import UNetConfig
    from wiw_amd.train_unet import EMAShadow

    params = {"a.weight": torch.randn(3, 5), "b.bias": torch.randn(7)}
    ema = EMAShadow(params)
    for _ in range(3):
        params["a.weight"].add_(1.0)
        ema.step(params)
    ucfg = C.unet_config_dict(UNetConfig.tiny(4))
    path = C.save_checkpoint(str(tmp_path), 3, params, {"exp_avg.a.weight": torch.zeros(3, 5)}, {"world": 1}, unet_config=ucfg,
                             ema={"shadow": ema.shadow, "state": ema.state()})
    with open(os.path.join(path, C.EMA_CONFIG_FILE)) as f:
        cfg = json.load(f)
    assert cfg["optimization_step"] == 3 and cfg["decay"] == 0.9999 and cfg["_class_name"] == ucfg["_class_name"]
    assert set(cfg) == set(ucfg) | set(C.EMA_STATE_KEYS)
    shadow, state = C.load_ema(path)
    other = EMAShadow({k: torch.zeros_like(v) for k, v in params.items()})
    other.load(shadow, state)
    assert other.optimization_step == 3 and all(torch.equal(other.shadow[k], ema.shadow[k]) for k in params)
    # the directory is a model directory: the serving loader reads its weights file
    from safetensors.torch import load_file
    assert set(load_file(os.path.join(path, C.EMA_FILE))) == set(params)


@pytest.mark.gpu
def test_ema_kernel_is_the_reference_expression_bit_for_bit():
    import wiw_amd  # noqa: F401
    from wiw_amd.hip import Hip
    from wiw_amd.train_unet import EMAShadow

    dev = torch.device("cuda:0")
    hip = Hip(dev)
    gen = torch.Generator().manual_seed(3)
    for n in (1, 3, 4, 1027, 320 * 320 * 9 + 2):
        s, p = torch.randn(n, generator=gen), torch.randn(n, generator=gen)
        for omd in (1.0, 1 - 2 / 11, 1e-4):
            ref = s.clone().sub_(omd * (s - p))
            out = s.clone().to(dev)
            hip.ema_step(out, p.to(dev), omd)
            assert torch.equal(out.cpu(), ref), (n, omd)
    # and the class on the device replays the reference fixture through the kernel
    g = np.load(GOLD)
    watch, frozen = [str(w) for w in g["watch"]], {str(f) for f in g["frozen"]}
    params = {w: torch.from_numpy(g[f"p0.{w}"]).clone().to(dev) for w in watch}
    _replay(EMAShadow(params, trainable=lambda n: n not in frozen, hip=hip), g, watch, params)


@pytest.mark.gpu
def test_trainer_with_ema_follows_the_parameters_and_resumes(tmp_path):
    """Trainer(use_ema=True): after every optimiser step the shadow moves by (1 - decay_t) towards the NEW parameters (train_svd.py:
    979-980); under `--train_param_type new` the frozen parameters' shadows stay copies; save -> load continues bit for bit."""
    import wiw_amd  # noqa: F401
    from wiw_amd import checkpoint as C
    from wiw_amd import train as T
    from wiw_amd.config import UNetConfig
    from wiw_amd.hip import Hip
    from wiw_amd.train_unet import Trainer, UNetTrain
    from wiw_amd.weights import random_state_dict

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "train_step_tiny.npz"))
    cfg = UNetConfig.tiny(4)
    hip = Hip(torch.device("cuda:0"))
    sd = random_state_dict(cfg, int(g["weight_seed"]))
    st = T.prepare_step(torch.from_numpy(g["latents"]), torch.from_numpy(g["noise"]), float(g["sigmas"][0]),
                        torch.from_numpy(g["conditional_latents"]), torch.from_numpy(g["encoder_hidden_states"]),
                        float(g["noise_aug_strength"]), torch.from_numpy(g["action_ids"]))
    a = Trainer(UNetTrain(cfg, sd, "cuda:0", hip=hip), lr=1e-3, use_ema=True, train_param_type="new")
    want = Trainer.PARAM_TYPES["new"]
    shadow = {k: v.clone() for k, v in a.net.master.items()}
    for step in (1, 2, 3):
        a.step(st)
        decay = 0.0 if step == 1 else (1 + step - 1) / (10 + step - 1)
        assert a.ema.optimization_step == step and a.ema.cur_decay_value == decay
        for k, p in a.net.master.items():
            shadow[k] = shadow[k] - (1 - decay) * (shadow[k] - p) if want(k) else p.clone()
            assert torch.equal(a.ema.shadow[k], shadow[k]), (step, k)
        if step == 2:
            path = a.save(str(tmp_path))
    moved = [k for k in sd if want(k) and not Trainer.is_dead(k) and not torch.equal(a.ema.shadow[k], a.net.master[k])]
    assert moved, "the averaged trainable parameters must lag behind the parameters"
    assert os.path.exists(os.path.join(path, C.EMA_FILE)) and os.path.exists(os.path.join(path, C.EMA_CONFIG_FILE))
    b = Trainer(UNetTrain(cfg, sd, "cuda:0", hip=hip), lr=1e-3, use_ema=True, train_param_type="new")
    b.load(path)
    assert b.ema.optimization_step == 2
    b.step(st)
    for k in sd:
        assert torch.equal(a.net.master[k], b.net.master[k]) and torch.equal(a.ema.shadow[k], b.ema.shadow[k]), k
