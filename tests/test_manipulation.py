"""task_type 'manipulation' — the reference worker's second served task (FTsvd/eval_inference.py:282-287, 324-331; the
manipulation planner speaks the same wire protocol, world-in-world-manip/wiw_manip/planner/igenex_planner.py:154-191).

`b_action` rows are continuous [x, y, z, qx, qy, qz, qw, gripper]; `get_action_ids` turns them into 10 channels
[norm_xyz | r6 | norm_grip] (utils/svd_utils.py:418-457, 499-567), the UNet embeds them with a 10-channel ActionEmbedder_
(or the 23-channel positional form, :570-592), and the initial noise is NOT rotated (pipeline:352-356).
tests/golden/manip_actions.npz holds what the reference's own functions return on seeded actions (oracle/make_golden.py
`manip`): the oracle restatement and the product's vectorised form are both checked against it; the GPU test runs the tiny
UNet with the 10-channel embedder against the reference module's output.
"""
import json
import os

import numpy as np
import pytest
import torch

import svd_oracle as O
import wiw_amd  # noqa: F401
from wiw_amd import pipeline as PL
from wiw_amd.server import plumbing as P
from wiw_amd.server.worker import SVDWorker, build_arg_parser


def test_oracle_and_product_action_ids_match_the_reference(golden):
    g = golden("manip_actions.npz")
    acts = g["actions"]
    assert acts.shape == (3, 14, 8)
    for fn in (O.manip_action_ids, PL.manip_action_ids):
        ab, re = fn(acts), fn(acts, absolute=False)
        assert ab.dtype == np.float32 and ab.shape == (3, 14, 10)
        # float64 arithmetic rounded to float32 once, as in the reference: equal up to the last float32 bit of a 2 pi range
        assert np.abs(ab - g["action_ids_abs"]).max() <= 1e-6, fn.__module__
        assert np.abs(re - g["action_ids_rel"]).max() <= 1e-6, fn.__module__
        assert np.all(re[:, 0] == 0)                       # no previous frame
    for fn in (O.action_encode_positional, PL.action_encode_positional):
        pos = fn(g["action_ids_abs"])
        assert pos.shape == (3, 14, 23) and np.array_equal(pos, g["action_ids_positional"])
    # the two restatements agree bit for bit with each other
    assert np.array_equal(O.manip_action_ids(acts), PL.manip_action_ids(acts))


def test_quaternion_matrix_properties():
    rs = np.random.RandomState(3)
    q = rs.standard_normal((64, 4)) * rs.uniform(0.1, 5.0, size=(64, 1))
    R = PL.quaternion_to_rotmatrix(q)
    assert np.abs(R @ np.swapaxes(R, -1, -2) - np.eye(3)).max() < 1e-12 and np.abs(np.linalg.det(R) - 1).max() < 1e-12
    assert np.abs(PL.quaternion_to_rotmatrix(-q) - R).max() < 1e-12          # q and -q are one rotation
    assert np.allclose(PL.quaternion_to_rotmatrix(np.array([0, 0, 0, 2.0])), np.eye(3))
    for i in range(4):
        assert np.abs(O.quaternion_to_rotmatrix(q[i]) - R[i]).max() < 1e-15


def test_value_ranges_follow_normalize_action():
    """xyz spans [-2 pi, 2 pi], r6 and the gripper [-pi, pi]; out-of-workspace positions and grips are clipped."""
    a = np.zeros((1, 3, 8), np.float32)
    a[0, :, 6] = 1.0                                          # identity rotation
    a[0, 0, :3] = [-0.3, -0.5, 0.6]; a[0, 1, :3] = [0.7, 0.5, 1.6]; a[0, 2, :3] = [5, 5, 5]
    a[0, :, 7] = [0.0, 1.0, 7.0]
    ids = PL.manip_action_ids(a)[0]
    assert np.allclose(ids[0, :3], -2 * np.pi, atol=1e-5) and np.allclose(ids[1, :3], 2 * np.pi, atol=1e-5)
    assert np.allclose(ids[2, :3], 2 * np.pi, atol=1e-5)      # clipped
    assert np.allclose(ids[:, 3:9], np.array([np.pi, 0, 0, np.pi, 0, 0]), atol=1e-5)     # r6 of I: (1,0,0,1,0,0) -> (pi,0,0,pi,0,0)
    assert np.allclose(ids[:, 9], [-np.pi, np.pi, np.pi], atol=1e-5)


class _Front:
    def encode(self, images, image_noise, noise_aug_strength):
        B, _, H, W = images.shape
        return np.zeros((B, 4, H // 8, W // 8), np.float32), np.zeros((B, 1, 1024), np.float32)

    def decode(self, latents):
        B, T, _, h, w = latents.shape
        return np.zeros((B, T, 3, 8 * h, 8 * w), np.float32)


def test_worker_accepts_manipulation_requests(tmp_path):
    seen = {}

    def denoise(il, ie, noise, actions, **kw):
        seen["actions"] = np.asarray(actions)
        return np.zeros_like(noise)

    w = SVDWorker(denoise, _Front(), width=128, height=64, out_width=48, out_height=40, task_type="manipulation")
    rs = np.random.RandomState(0)
    acts = rs.standard_normal((2, 14, 8)).astype(np.float32)
    req = {"b_action": acts.tolist(), "save_dirs": [str(tmp_path / f"m{i}") for i in range(2)], "request_model_name": "igen",
           "b_image": rs.randint(0, 256, size=(2, 3, 64, 128), dtype=np.uint8)}
    P.validate_request(req, 14, "manipulation")
    out = w(req)                                               # file mode: frames + action_seq.json per candidate
    assert seen["actions"].shape == (2, 14, 8) and out["save_dirs"] == req["save_dirs"]
    with open(os.path.join(req["save_dirs"][1], "action_seq.json")) as f:
        assert np.allclose(np.array(json.load(f)), acts[1], atol=1e-6)
    # the shape contract is per task type
    with pytest.raises(AssertionError):
        P.validate_request(dict(req, b_action=np.zeros((2, 14), np.int64)), 14, "manipulation")
    with pytest.raises(AssertionError):
        SVDWorker(denoise, _Front(), width=128, height=64)(req)          # a navigation worker refuses 3-D actions
    args = build_arg_parser().parse_args(["--task_type", "manipulation", "--action_input_channel", "10"])
    assert args.task_type == "manipulation" and args.action_input_channel == 10


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_unet_with_manipulation_actions_on_the_hip_path(dtype, golden):
    """Tiny UNet, task_type='manipulation', 10-channel ActionEmbedder_ (K = 120 padded to 128), against the reference
    module's fp32 output; then the loop: 3-D actions -> ids inside `SVDDenoiser.denoise`, no noise rotation."""
    from wiw_amd.config import UNetConfig
    from wiw_amd.pipeline import SVDDenoiser
    from wiw_amd.unet import UNetHIP
    from wiw_amd.weights import random_state_dict

    g = golden("manip_actions.npz")
    cfg = UNetConfig(block_out_channels=(64, 128, 128, 128), num_attention_heads=(1, 2, 2, 2), num_frames=4,
                     action_input_channel=10, task_type="manipulation")
    sd = random_state_dict(cfg, int(g["unet_weight_seed"]))
    unet = UNetHIP(cfg, sd, "cuda:0", dtype=dtype)
    out = unet(torch.from_numpy(g["unet_sample"]), float(g["unet_timestep"]), torch.from_numpy(g["unet_ehs"]),
               torch.from_numpy(g["unet_added_time_ids"]), torch.from_numpy(g["unet_action_ids"])).cpu().numpy()
    ref = g["unet_out"]
    rms = float(np.sqrt(((out - ref) ** 2).mean()) / np.sqrt((ref ** 2).mean()))
    print(f"[parity] tiny UNet, manipulation actions (10 channels), {dtype}: rms_rel={rms:.3e}")
    assert np.isfinite(out).all() and rms <= (1.8e-2 if dtype == torch.bfloat16 else 2.3e-3)     # 1.25 x measured (1.44e-2 / 1.82e-3)
    # the loop: 3-D actions, ids computed inside, the noise is NOT rotated (pipeline:352-356) — vs the oracle loop on the ids
    rs = np.random.RandomState(2)
    il = torch.from_numpy(rs.standard_normal((1, 4, 16, 32)).astype(np.float32))
    ie = torch.from_numpy(rs.standard_normal((1, 1, cfg.cross_attention_dim)).astype(np.float32))
    noise = torch.from_numpy(rs.standard_normal((1, 4, 4, 16, 32)).astype(np.float32))
    acts = g["actions"][:1, :4]
    lat = SVDDenoiser(unet).denoise(il, ie, noise, acts, num_steps=2).cpu()
    sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
    with torch.no_grad():
        ref = O.denoise(sdt, cfg.as_dict(), il, ie, noise, acts, num_steps=2)
    rms = float((lat - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    print(f"[parity] 2-step loop, manipulation actions, {dtype}: rms_rel={rms:.3e}")
    assert rms <= (2.2e-2 if dtype == torch.bfloat16 else 2.7e-3)       # 1.25 x measured (1.72e-2 / 2.15e-3)
