"""Block / UNet / denoise-loop parity of the HIP path against the golden vectors produced by the
reference (tests/golden, oracle/make_golden.py) and against the CPU oracle.  GPU only.

Tolerance: the HIP path computes in bf16 (fp32 accumulate / statistics).  The fixtures carry the
reference's OWN bf16 run (`out_ref_bf16`) next to its fp32 run, so "bf16-class error" is measured,
not guessed: gate = HIP error vs the fp32 reference <= the reference's OWN bf16-vs-fp32 error on
the same inputs (the HIP path must be at least as accurate as the reference run in bf16).
Measured on MI355X (round 1): UNet forward rms 1.5e-2 (reference bf16: 2.0e-2); 3-step loop rms
2.7e-2 (reference bf16: 3.5e-2).  north_star's 1e-3 is not reachable by ANY bf16 evaluation of this
network, the reference's included — see DESIGN.md §numerics."""
import numpy as np
import pytest
import torch

import svd_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30)), float(
        np.sqrt(((a - b) ** 2).mean()) / (np.sqrt((b ** 2).mean()) + 1e-30))


@pytest.fixture(scope="module")
def tiny():
    import wiw_amd  # noqa: F401
    from wiw_amd.config import UNetConfig
    from wiw_amd.unet import UNetHIP
    from wiw_amd.weights import random_state_dict

    cache = {}

    def get(seed):
        if seed not in cache:
            cfg = UNetConfig.tiny(4)
            sd = random_state_dict(cfg, seed)
            cache[seed] = (cfg, sd, UNetHIP(cfg, sd, DEV))
        return cache[seed]

    return get


def tok(x):  # (N,C,H,W) fp32 -> token-major bf16 on device
    return torch.from_numpy(np.ascontiguousarray(x)).permute(0, 2, 3, 1).reshape(-1, x.shape[1]).to(DEV, torch.bfloat16).contiguous()


def untok(t, n, h, w):
    return t.float().cpu().reshape(n, h, w, -1).permute(0, 3, 1, 2).numpy()


def test_res_block_golden(tiny, golden):
    g = golden("blocks_tiny.npz")
    cfg, sd, unet = tiny(int(g["weight_seed"]))
    p = str(g["res_prefix"])
    x, temb, ref = g["res_in"], g["res_temb"], g["res_out"]
    n, Cin, h, w = x.shape
    Cout = ref.shape[1]
    C1 = Cin - Cout  # up-block input = [hidden | skip] concat (unet_3d_blocks.py:1612)
    # time-embedding projections through the batched GEMM, as in forward()
    emb = torch.nn.functional.silu(torch.from_numpy(temb)).to(DEV, torch.bfloat16).contiguous()
    temb_all = torch.empty(n, unet.temb_total, dtype=torch.float32, device=DEV)
    unet.hip.gemm(emb, unet.w["temb_all.weight"], temb_all, M=n, N=unet.temb_total, K=cfg.time_embed_dim,
                  C1=cfg.time_embed_dim, bias=unet.w["temb_all.bias"], epilogue=4)
    out = unet._res_block(p, tok(x[:, :C1]), C1, tok(x[:, C1:]), Cin - C1, Cout, n * h * w, h, w, temb_all, float(g["res_eps"]))
    mx, rms = rel(untok(out, n, h, w), ref)
    print(f"[parity] res block {p}: max_rel={mx:.3e} rms_rel={rms:.3e}")
    assert mx < 2.5e-2 and rms < 1e-2


def test_transformer_golden(tiny, golden):
    g = golden("blocks_tiny.npz")
    cfg, sd, unet = tiny(int(g["weight_seed"]))
    p = str(g["tr_prefix"])
    x, ehs, ref = g["tr_in"], g["tr_ehs"], g["tr_out"]
    n, Cn, h, w = x.shape
    T = cfg.num_frames
    Bc = n // T
    ehs_b = torch.from_numpy(ehs).reshape(Bc, T, 1, -1)[:, 0]  # one token per CFG item
    assert float(ehs_b[0].abs().max()) == 0.0
    cond = unet.prepare_request(ehs_b[Bc // 2:], np.zeros((Bc // 2, T, T), np.float32))
    out = unet._transformer(p, tok(x), Cn, n * h * w, h, w, int(g["tr_heads"]), cond)
    mx, rms = rel(untok(out, n, h, w), ref)
    print(f"[parity] transformer {p}: max_rel={mx:.3e} rms_rel={rms:.3e}")
    assert mx < 2.5e-2 and rms < 1e-2


def _run_unet(unet, g):
    return unet(torch.from_numpy(g["sample"]), float(g["timestep"]), torch.from_numpy(g["ehs"]),
                torch.from_numpy(g["added_time_ids"]), torch.from_numpy(g["action_ids"])).cpu().numpy()


def test_unet_tiny_b1_golden(tiny, golden):
    g = golden("unet_tiny_b1.npz")
    cfg, sd, unet = tiny(int(g["weight_seed"]))
    out = _run_unet(unet, g)
    mx, rms = rel(out, g["out"])
    mx_ref, rms_ref = rel(g["out_ref_bf16"], g["out"])
    print(f"[parity] unet tiny B=1: HIP max_rel={mx:.3e} rms_rel={rms:.3e} | reference bf16 run: {mx_ref:.3e} {rms_ref:.3e}")
    assert np.isfinite(out).all()
    assert rms <= rms_ref and mx <= 1.25 * mx_ref


def test_unet_tiny_b2_contract(tiny, golden):
    """Batch 2 must equal two independent reference B=1 runs (NOT the reference's cross-wired batch)."""
    g = golden("unet_tiny_b2.npz")
    g1 = golden("unet_tiny_b1.npz")  # same weights: its reference-bf16 error is the yardstick
    mx_ref, rms_ref = rel(g1["out_ref_bf16"], g1["out"])
    cfg, sd, unet = tiny(int(g["weight_seed"]))
    out = _run_unet(unet, g)
    mx, rms = rel(out, g["out_contract"])
    mxq, rmsq = rel(out, g["out_reference_batched"])
    print(f"[parity] unet tiny B=2: vs contract max_rel={mx:.3e} rms_rel={rms:.3e}; vs cross-wired batch rms={rmsq:.3e}")
    assert rms <= rms_ref and mx <= 1.25 * mx_ref
    assert rmsq > 2 * rms  # demonstrably not reproducing the §9.2 defect


def test_denoise_loop_golden(tiny, golden):
    g = golden("pipeline_tiny.npz")
    from wiw_amd.pipeline import SVDDenoiser

    cfg, sd, unet = tiny(int(g["weight_seed"]))
    den = SVDDenoiser(unet)
    lat = den.denoise(torch.from_numpy(g["image_latents"]), torch.from_numpy(g["image_embeddings"]),
                      torch.from_numpy(g["latent_noise"]), g["actions"], num_steps=int(g["num_steps"]))
    mx, rms = rel(lat.cpu().numpy(), g["latents_out"])
    mx_ref, rms_ref = rel(g["latents_out_ref_bf16"], g["latents_out"])
    print(f"[parity] 3-step denoise loop (B=2): max_rel={mx:.3e} rms_rel={rms:.3e} | reference bf16 run: {mx_ref:.3e} {rms_ref:.3e}")
    assert rms <= rms_ref and mx <= 1.25 * mx_ref


def test_unet_vs_oracle_other_seed(tiny):
    """Same check against the CPU oracle (not a stored vector): different weights / inputs / frames-first quirks."""
    cfg, sd, unet = tiny(11)
    rs = np.random.RandomState(3)
    B, T, h, w = 1, cfg.num_frames, 8, 64
    sample = rs.standard_normal((2 * B, T, 8, h, w)).astype(np.float32)
    sample[:B, :, 4:] = 0
    ehs = rs.standard_normal((2 * B, 1, cfg.cross_attention_dim)).astype(np.float32)
    ehs[:B] = 0
    tids = np.tile(np.array([[6, 127, 0.02]], np.float32), (2 * B, 1))
    aid = O.action_ids_idx_encode(np.array([[4, 3, 1, 2]]))
    ref = O.unet_forward({k: torch.from_numpy(v) for k, v in sd.items()}, cfg.as_dict(), torch.from_numpy(sample),
                         torch.tensor(-0.75), torch.from_numpy(ehs), torch.from_numpy(tids), torch.from_numpy(aid)).numpy()
    out = unet(torch.from_numpy(sample), -0.75, torch.from_numpy(ehs), torch.from_numpy(tids), torch.from_numpy(aid)).cpu().numpy()
    mx, rms = rel(out, ref)
    print(f"[parity] unet tiny vs oracle (seed 11, 8x64): max_rel={mx:.3e} rms_rel={rms:.3e}")
    assert rms < 2e-2 and mx < 4e-2


def test_unet_with_folded_layernorms(tiny, golden):
    """UNetHIP(fold_layernorm=True): norm1 / norm3 / norm_in evaluated INSIDE their consumer GEMMs (WIW_EPI_LNFOLD, row
    statistics from the operand fragments).  Same gate as the default network, and close to it."""
    from wiw_amd.unet import UNetHIP

    g = golden("unet_tiny_b1.npz")
    cfg, sd, unet = tiny(int(g["weight_seed"]))
    folded = UNetHIP(cfg, sd, DEV, hip=unet.hip, fold_layernorm=True)
    assert any(k.endswith(".lnfold.weight") for k in folded.w) and not any(k.endswith(".lnfold.weight") for k in unet.w)
    out, base = _run_unet(folded, g), _run_unet(unet, g)
    mx, rms = rel(out, g["out"])
    mx_ref, rms_ref = rel(g["out_ref_bf16"], g["out"])
    print(f"[parity] unet tiny, folded LayerNorms: max_rel={mx:.3e} rms_rel={rms:.3e} (reference bf16 run {rms_ref:.3e}); "
          f"vs the unfolded network rms={rel(out, base)[1]:.3e}")
    assert np.isfinite(out).all() and rms <= rms_ref and mx <= 1.25 * mx_ref


def test_more_conditioning_embeddings_than_the_replica_was_built_for_are_refused(tiny):
    """The served cross-attention is the closed form for ONE key; (B, P > 1, D) embeddings of a --num_past_obs > 1 checkpoint
    need a replica built with num_past_obs = P (the general attention, below) — a replica built for one token must raise, not
    flatten P tokens into a mis-shaped vector (pipeline_stable_video_diffusion.py:501-504)."""
    cfg, sd, unet = tiny(0)
    ie = torch.zeros(1, 2, cfg.cross_attention_dim)
    with pytest.raises(NotImplementedError, match="num_past_obs"):
        unet.prepare_request(ie, np.zeros((1, cfg.num_frames, cfg.action_input_channel), np.float32))
    sample = torch.zeros(2, cfg.num_frames, 8, 16, 32)
    with pytest.raises(NotImplementedError, match="num_past_obs"):
        unet(sample, 1.0, torch.zeros(2, 2, cfg.cross_attention_dim), torch.zeros(2, 3), torch.zeros(1, cfg.num_frames, cfg.action_input_channel))


@pytest.mark.parametrize("name", ["bf16", "fp16", "fp16+res32"])
def test_unet_with_three_conditioning_tokens_against_the_reference(name, golden):
    """--num_past_obs > 1 (train_svd.py:359, 889-894): encoder_hidden_states (2, 3, 1024), i.e. Sk = 3 in both cross-attentions of
    every transformer layer (attention.py:545-551, 740-743) — `UNetHIP(num_past_obs=3)` on `wiw_cross_attn_fewkeys_bf16` against
    the REFERENCE's forward (tests/golden/unet_tiny_ctx3.npz).  A replica built for three tokens also serves one (same bytes
    as the closed form of a one-token replica), and the three-token result is NOT what the first token alone gives."""
    import wiw_amd  # noqa: F401
    from wiw_amd.config import UNetConfig
    from wiw_amd.hip import Hip
    from wiw_amd.unet import UNetHIP
    from wiw_amd.weights import random_state_dict

    g = golden("unet_tiny_ctx3.npz")
    g1 = golden("unet_tiny_b1.npz")
    cfg = UNetConfig.tiny(4)
    sd = random_state_dict(cfg, int(g["weight_seed"]))
    dt = torch.bfloat16 if name == "bf16" else torch.float16
    hip = Hip(torch.device(DEV), dt)
    unet3 = UNetHIP(cfg, sd, DEV, hip=hip, num_past_obs=3, residual_fp32=name.endswith("res32"))
    out = _run_unet(unet3, g)
    mx, rms = rel(out, g["out"])
    mx_ref, rms_ref = rel(g1["out_ref_bf16"], g1["out"])          # the reference's own bf16 run of this tiny network: the yardstick
    far = rel(g["out_first_token_only"], g["out"])[1]
    print(f"[parity] unet tiny, 3 conditioning tokens ({name}): max_rel={mx:.3e} rms_rel={rms:.3e} | reference bf16 yardstick "
          f"{mx_ref:.3e} {rms_ref:.3e} | first token alone is {far:.3e} away")
    assert np.isfinite(out).all() and rms <= rms_ref and mx <= 1.25 * mx_ref
    assert far > 4 * rms, "the fixture must tell one token from three"
    if name == "bf16":
        unet1 = UNetHIP(cfg, sd, DEV, hip=hip)
        a, b = _run_unet(unet1, g1), _run_unet(unet3, g1)
        assert np.array_equal(a, b), "one token on a replica built for three: the closed form, same bytes"
        with pytest.raises(NotImplementedError, match="num_past_obs"):
            unet3(torch.zeros(2, cfg.num_frames, 8, 16, 32), 1.0, torch.zeros(2, 4, cfg.cross_attention_dim), torch.zeros(2, 3),
                  torch.zeros(1, cfg.num_frames, cfg.action_input_channel))
