"""hipGraph replay of the UNet forward (pipeline.GraphedForward) against the eager loop: the same bytes.  GPU only.

The C ABI allocates and synchronises nothing inside its entry points (include/wiw_svd.h conventions, SURVEY 8b "graph-capture
safe"): this test is the proof — one forward (~1 100 launches at full size) is captured once per (candidates, h, w) and
replayed for every Euler step and every later request of that shape."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _inputs(cfg, B, T, h, w, seed):
    rs = np.random.RandomState(seed)
    il = torch.from_numpy(rs.standard_normal((B, 4, h, w)).astype(np.float32))
    ie = torch.from_numpy(rs.standard_normal((B, 1, cfg.cross_attention_dim)).astype(np.float32))
    noise = torch.from_numpy(rs.standard_normal((B, T, 4, h, w)).astype(np.float32))
    acts = np.stack([np.roll(np.array([4, 2, 1, 3][:T]), b) for b in range(B)])
    acts[:, 0] = 4
    return il, ie, noise, acts


def test_graph_replay_is_bit_identical_and_reused_across_requests():
    import wiw_amd  # noqa: F401
    from wiw_amd.config import UNetConfig
    from wiw_amd.pipeline import SVDDenoiser
    from wiw_amd.unet import UNetHIP
    from wiw_amd.weights import random_state_dict

    cfg = UNetConfig.tiny(4)
    unet = UNetHIP(cfg, random_state_dict(cfg, 5), DEV)
    eager, graphed = SVDDenoiser(unet, use_graph=False), SVDDenoiser(unet, use_graph=True)
    T, h, w = 4, 16, 32
    for B, seed in ((1, 0), (1, 1), (2, 2), (1, 3)):        # second and fourth request reuse the B = 1 graph
        il, ie, noise, acts = _inputs(cfg, B, T, h, w, seed)
        a = eager.denoise(il, ie, noise, acts, num_steps=3)
        b = graphed.denoise(il, ie, noise, acts, num_steps=3)
        assert graphed.graph_error is None, graphed.graph_error
        assert torch.equal(a, b), f"graph replay differs from the eager loop (B={B}, seed={seed})"
    assert set(graphed._graphs) == {(1, h, w), (2, h, w)}
    assert graphed.host_launch["graph"][1] == 12 and graphed.host_launch["eager"][1] == 0
    assert eager.host_launch["eager"][1] == 12


def test_eager_steps_interleave_with_replays():
    """bench.py records per-launch events on every n-th Euler step: those steps run eagerly on the graph's static buffers."""
    import wiw_amd  # noqa: F401
    from wiw_amd.config import UNetConfig
    from wiw_amd.pipeline import SVDDenoiser
    from wiw_amd.unet import UNetHIP
    from wiw_amd.weights import random_state_dict

    cfg = UNetConfig.tiny(4)
    unet = UNetHIP(cfg, random_state_dict(cfg, 6), DEV)
    il, ie, noise, acts = _inputs(cfg, 1, 4, 16, 32, 7)
    ref = SVDDenoiser(unet, use_graph=False).denoise(il, ie, noise, acts, num_steps=4)
    den = SVDDenoiser(unet, use_graph=True)
    prof = []

    def toggle(i, _lat):     # after step i: arm events for odd steps
        unet.hip.gemm_profile = prof if (i + 1) % 2 == 1 else None
    out = den.denoise(il, ie, noise, acts, num_steps=4, callback=toggle)
    unet.hip.gemm_profile = None
    assert torch.equal(out, ref)
    assert den.host_launch["graph"][1] == 2 and den.host_launch["eager"][1] == 2 and len(prof) > 0


def test_capture_is_refused_loudly_when_the_pool_would_not_fit(caplog):
    """a capture pins ~6 GB per candidate at 576x1024; the denoiser estimates the pool against the free HBM, refuses
    the SHAPE (not graphs for good) with a logged warning, runs it eagerly with the same bytes, and retries after an eviction."""
    import logging

    import wiw_amd  # noqa: F401
    from wiw_amd.config import UNetConfig
    from wiw_amd.pipeline import SVDDenoiser
    from wiw_amd.unet import UNetHIP
    from wiw_amd.weights import random_state_dict

    cfg = UNetConfig.tiny(4)
    unet = UNetHIP(cfg, random_state_dict(cfg, 5), DEV)
    den = SVDDenoiser(unet, use_graph=True)
    il, ie, noise, acts = _inputs(cfg, 1, 4, 16, 32, 0)
    ref = SVDDenoiser(unet, use_graph=False).denoise(il, ie, noise, acts, num_steps=2)
    den.GRAPH_MEM_FRACTION = 0.0                      # nothing fits
    with caplog.at_level(logging.WARNING, logger="wiw_amd.graph"):
        out = den.denoise(il, ie, noise, acts, num_steps=2)
    assert torch.equal(out, ref) and den.host_launch["eager"][1] == 2 and not den._graphs
    assert "MemoryError" in den.graph_error and any("EAGERLY" in r.message for r in caplog.records)
    st = den.graph_status()
    assert st["enabled"] and st["captured"] == [] and len(st["refused"]) == 1
    den.GRAPH_MEM_FRACTION = 0.8                      # a different shape still captures; the refused one stays refused ...
    il2, ie2, noise2, acts2 = _inputs(cfg, 2, 4, 16, 32, 1)
    den.denoise(il2, ie2, noise2, acts2, num_steps=2)
    assert list(den._graphs) == [(2, 16, 32)] and (1, 16, 32) in den._graph_refused
    den.MAX_GRAPHS = 1                                # ... until an eviction gives memory back
    il3, ie3, noise3, acts3 = _inputs(cfg, 3, 4, 16, 32, 2)
    den.denoise(il3, ie3, noise3, acts3, num_steps=2)
    assert not den._graph_refused
    out = den.denoise(il, ie, noise, acts, num_steps=2)
    assert torch.equal(out, ref) and list(den._graphs) == [(1, 16, 32)]
    assert unet.hip.gn_counters_clean()               # every GroupNorm statistics launch (eager, warm-up, captured) left zeros


def test_many_shapes_captured_in_sequence_keep_capturing():
    """GroupNorm counter rows are keyed by stream; every capture used to take a FRESH side stream for its warm-up,
    so the 15th shape a server saw ran out of rows, was refused (after evicting a live graph) and the server stayed eager.
    One persistent warm-up stream per Hip: 20 shapes in sequence all capture, three streams ever launch a GroupNorm."""
    import wiw_amd  # noqa: F401
    from wiw_amd.config import UNetConfig
    from wiw_amd.pipeline import SVDDenoiser
    from wiw_amd.unet import UNetHIP
    from wiw_amd.weights import random_state_dict

    cfg = UNetConfig.tiny(4)
    unet = UNetHIP(cfg, random_state_dict(cfg, 5), DEV)
    den = SVDDenoiser(unet, use_graph=True)
    shapes = [(1 + (i % 3), 16 * (1 + i // 6), 32 * (1 + (i // 3) % 2)) for i in range(20)]
    assert len(set(shapes)) == 20
    for i, (B, h, w) in enumerate(shapes):
        il, ie, noise, acts = _inputs(cfg, B, 4, h, w, i)
        den.denoise(il, ie, noise, acts, num_steps=1)
        assert den.graph_error is None, f"shape {i} {B, h, w}: {den.graph_error}"
        assert (B, h, w) in den._graphs and len(den._graphs) <= den.MAX_GRAPHS
    assert den.host_launch["eager"][1] == 0 and den.host_launch["graph"][1] == 20
    assert len(unet.hip._gn_cnt) <= 3, f"GroupNorm launched from {len(unet.hip._gn_cnt)} streams"
    assert unet.hip.gn_counters_clean()


def test_a_shape_that_cannot_fit_costs_no_live_graph_and_its_refusal_expires():
    """the fit is checked BEFORE the eviction (counting what the eviction would return), and a refusal is not
    for ever — it expires after REFUSAL_TTL requests of that shape."""
    import wiw_amd  # noqa: F401
    from wiw_amd.config import UNetConfig
    from wiw_amd.pipeline import SVDDenoiser
    from wiw_amd.unet import UNetHIP
    from wiw_amd.weights import random_state_dict

    cfg = UNetConfig.tiny(4)
    unet = UNetHIP(cfg, random_state_dict(cfg, 5), DEV)
    den = SVDDenoiser(unet, use_graph=True)
    den.MAX_GRAPHS = 1
    a = _inputs(cfg, 1, 4, 16, 32, 0)
    b = _inputs(cfg, 2, 4, 16, 32, 1)
    den.denoise(*a, num_steps=1)
    assert list(den._graphs) == [(1, 16, 32)]
    est = den.graph_pool_estimate
    den.graph_pool_estimate = lambda B, h, w: 1e18 if B == 2 else est(B, h, w)     # B = 2 can never fit
    den.denoise(*b, num_steps=1)
    assert list(den._graphs) == [(1, 16, 32)], "a shape that cannot fit must not evict a live graph"
    assert (2, 16, 32) in den._graph_refused
    den.graph_pool_estimate = est
    for _ in range(den.REFUSAL_TTL - 1):                                            # still refused: runs eagerly
        den.denoise(*b, num_steps=1)
    assert (2, 16, 32) in den._graph_refused or (2, 16, 32) in den._graphs
    den.denoise(*b, num_steps=1)
    assert list(den._graphs) == [(2, 16, 32)] and not den._graph_refused
