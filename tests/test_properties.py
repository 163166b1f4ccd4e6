"""Property tests (hypothesis) of the host-side logic: wire framing, request split / recompose, candidate sharding,
action encoding and the noise-rotation rule — the size-independent invariants of the boundary (CPU only)."""
import io
import pickle
import struct

import numpy as np
import torch
from hypothesis import given, settings, strategies as st

import svd_oracle as O
import wiw_amd  # noqa: F401
from wiw_amd import pipeline as PL
from wiw_amd.parallel import ShardedWorker, shard_bounds
from wiw_amd.server import plumbing as P
from wiw_amd.server import protocol as PR


class _Sock:
    """Minimal socket stand-in: sendall appends, recv pops (any chunking)."""

    def __init__(self, chunk):
        self.buf, self.chunk = io.BytesIO(), chunk

    def sendall(self, b):
        pos = self.buf.tell()
        self.buf.seek(0, 2)
        self.buf.write(b)
        self.buf.seek(pos)

    def recv(self, n):
        return self.buf.read(min(n, self.chunk))


@settings(max_examples=40, deadline=None)
@given(st.integers(1, 9), st.integers(1, 4096), st.integers(0, 2 ** 31 - 1))
def test_frames_survive_any_chunking(n_msgs, chunk, seed):
    """`>I` length + pickle framing (worker_manager.py:274-287): messages come back whole and in order however the
    transport fragments them."""
    rs = np.random.RandomState(seed)
    msgs = [{"k": rs.randint(0, 255, size=rs.randint(0, 3000), dtype=np.uint8), "i": i, "s": "x" * int(rs.randint(0, 50))}
            for i in range(n_msgs)]
    s = _Sock(chunk)
    for m in msgs:
        PR.write_framed(s, m)
    raw = s.buf.getvalue()
    assert struct.unpack(">I", raw[:4])[0] == len(pickle.dumps(msgs[0], protocol=pickle.HIGHEST_PROTOCOL))
    for m in msgs:
        got = PR.read_framed(s)
        assert got["i"] == m["i"] and got["s"] == m["s"] and np.array_equal(got["k"], m["k"])


@settings(max_examples=60, deadline=None)
@given(st.integers(1, 40), st.integers(1, 12))
def test_split_then_recompose_is_identity_on_lists(b, bs):
    """Batcher.split_batch / _recompose_batch (worker_manager.py:448-481)."""
    req = {"save_dirs": [f"d{i}" for i in range(b)], "b_action": [[i] * 14 for i in range(b)]}
    parts = P.split_batch(req, bs)
    assert sum(len(p["save_dirs"]) for p in parts) == b and all(0 < len(p["save_dirs"]) <= bs for p in parts)
    assert P.recompose(parts) == req


@settings(max_examples=100, deadline=None)
@given(st.integers(0, 500), st.integers(1, 16))
def test_shard_bounds_partition(n, world):
    b = shard_bounds(n, world)
    assert len(b) == world and b[0][0] == 0 and b[-1][1] == n
    assert all(lo <= hi for lo, hi in b) and all(b[k][1] == b[k + 1][0] for k in range(world - 1))
    sizes = [hi - lo for lo, hi in b]
    assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


@settings(max_examples=30, deadline=None)
@given(st.integers(1, 9), st.integers(1, 5), st.integers(0, 10 ** 6))
def test_sharded_worker_concat_is_order_preserving(b, parts, seed):
    rs = np.random.RandomState(seed)
    frames = rs.randint(0, 255, size=(b, 2, 3), dtype=np.uint8)
    dirs = [f"c{i}" for i in range(b)]
    import torch

    from wiw_amd.parallel import pad_slices, unpad_concat
    bounds = shard_bounds(b, parts)
    width = max(hi - lo for lo, hi in bounds)
    padded = pad_slices(torch.from_numpy(frames), bounds, width)      # what scatter sends / gather receives
    assert len(padded) == parts and all(p.shape == (width, 2, 3) for p in padded)
    assert np.array_equal(unpad_concat(padded, bounds).numpy(), frames)
    assert [d for lo, hi in bounds for d in dirs[lo:hi]] == dirs
    sl = ShardedWorker._slice({"request_model_name": "igen", "save_dirs": dirs, "b_image": frames}, 0, 1)
    assert sl["request_model_name"] == "igen" and sl["save_dirs"] == dirs[:1] and sl["b_image"].shape[0] == 1


@settings(max_examples=60, deadline=None)
@given(st.lists(st.lists(st.integers(0, 4), min_size=14, max_size=14), min_size=1, max_size=5))
def test_action_encoding_matches_oracle_and_is_cumulative(rows):
    """action_ids_idx_encode (svd_utils.py:594-632): row i = [stop, a_1..a_i, 0...]; first action forced to 'stop' (4)."""
    a = np.asarray(rows, dtype=np.int64)
    enc = PL.action_ids_idx_encode(a)
    assert np.array_equal(enc, O.action_ids_idx_encode(a))
    assert enc.shape == (a.shape[0], 14, 14) and (enc[:, :, 0] == 4).all()
    for i in range(14):
        assert (enc[:, i, i + 1:] == 0).all() and np.array_equal(enc[:, i, 1:i + 1], a[:, 1:i + 1])


@settings(max_examples=40, deadline=None)
@given(st.lists(st.integers(1, 4), min_size=5, max_size=5), st.integers(0, 10 ** 6))
def test_noise_rotation_is_a_roll_chain(acts, seed):
    """sample_latent_noise / rotate_by_degrees (pipeline:750-823): a turn at frame i makes noise[i] a horizontal roll of
    noise[i-1] by W/16 — values are permuted, never changed, and the oracle agrees bit for bit."""
    g = torch.Generator().manual_seed(seed)
    noise = torch.randn(1, 6, 4, 4, 32, generator=g)
    a = np.asarray([[4] + acts], dtype=np.int64)
    out = PL.rotate_latent_noise(noise.clone(), a)
    assert torch.equal(out, O.rotate_latent_noise(noise.clone(), a))
    for i in range(1, 6):
        if a[0, i] in (2, 3):
            shift = 32 // 16 * (1 if a[0, i] == 2 else -1)
            fwd = torch.equal(out[0, i], torch.roll(out[0, i - 1], shift, dims=-1))
            assert fwd or torch.equal(out[0, i], torch.roll(out[0, i - 1], -shift, dims=-1))
        else:
            assert torch.equal(out[0, i], noise[0, i])


@settings(max_examples=50, deadline=None)
@given(st.integers(2, 80))
def test_scheduler_tables_for_any_step_count(n):
    """Karras rho=7 table (scheduler:476-499): product host code == oracle bit for bit, strictly decreasing from 700 to
    0.002, terminal 0; continuous timesteps t = 0.25 ln sigma; init_noise_sigma = sqrt(sigma_max^2 + 1)."""
    s, so = PL.karras_sigmas(n), O.karras_sigmas(n)
    assert s.dtype == np.float32 and np.array_equal(s, so) and len(s) == n + 1
    assert s[0] == np.float32(700.0) and abs(float(s[-2]) - 0.002) < 1e-6 and s[-1] == 0.0
    assert (np.diff(s) < 0).all()
    assert np.array_equal(PL.sigma_to_timestep(s), O.sigma_to_timestep(so))
    assert PL.init_noise_sigma(s) == O.init_noise_sigma(so) == float((700.0 ** 2 + 1) ** 0.5)


@settings(max_examples=40, deadline=None)
@given(st.floats(0.01, 700.0), st.floats(0.0, 1.0), st.integers(0, 10 ** 6))
def test_euler_step_is_affine_in_latents_and_prediction(sigma, frac, seed):
    """EulerDiscreteScheduler.step with v-prediction (scheduler:635-673) is affine: step(a v1 + b v2, a x1 + b x2) =
    a step(v1, x1) + b step(v2, x2) for a + b = 1 — and sigma_next = sigma is the identity."""
    g = torch.Generator().manual_seed(seed)
    v1, v2, x1, x2 = (torch.randn(2, 3, 4, generator=g, dtype=torch.float64) for _ in range(4))
    sn = sigma * frac
    a, b = 0.3, 0.7
    lhs = O.euler_step(a * v1 + b * v2, a * x1 + b * x2, sigma, sn)
    rhs = a * O.euler_step(v1, x1, sigma, sn) + b * O.euler_step(v2, x2, sigma, sn)
    assert torch.allclose(lhs.double(), rhs.double(), rtol=1e-4, atol=1e-4)
    assert torch.allclose(O.euler_step(v1, x1, sigma, sigma).double(), x1, rtol=1e-5, atol=1e-5)
