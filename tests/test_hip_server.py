"""End-to-end on the GPU: wire request -> SVDWorker (PyTorch frontend + HIP denoiser) -> uint8 response,
compared with the same chain evaluated by the CPU oracle (fp32) on identical noise draws."""
import numpy as np
import pytest
import torch

import svd_oracle as O
import vae_oracle as VO

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,psnr_gate", [(torch.bfloat16, 30.0), (torch.float16, 45.0)], ids=["bf16", "fp16"])
def test_worker_end_to_end_matches_oracle_chain(tmp_path, dtype, psnr_gate):
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection

    import wiw_amd  # noqa: F401
    from wiw_amd import frontend as FE
    from wiw_amd.config import UNetConfig
    from wiw_amd.pipeline import SVDDenoiser
    from wiw_amd.server import plumbing as P
    from wiw_amd.server.worker import SVDWorker
    from wiw_amd.unet import UNetHIP
    from wiw_amd.weights import random_state_dict

    T, H, W = 4, 128, 256
    cfg = UNetConfig.tiny(T)
    sd = random_state_dict(cfg, 21)
    vcfg = dict(block_out_channels=(32, 64, 64, 64), layers_per_block=1)
    vsd = FE.vae_random_state_dict(22, **vcfg)
    torch.manual_seed(0)
    clip = CLIPVisionModelWithProjection(CLIPVisionConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=1,
                                                          num_attention_heads=2, image_size=224, patch_size=32,
                                                          projection_dim=1024)).eval()
    den = SVDDenoiser(UNetHIP(cfg, sd, "cuda:0", dtype=dtype))   # bf16 (BASELINE) / fp16 (the reference's served default)
    fe_gpu = VO.TorchFrontend(vsd, clip, device="cuda:0", vae_dtype=torch.float32, dtype=torch.float32, **vcfg)

    def denoise(il, ie, nz, act, **kw):
        return den.denoise(torch.from_numpy(il), torch.from_numpy(ie), torch.from_numpy(nz), act, **kw).cpu().numpy()

    draws = []

    def noise_fn(shape):
        rs = np.random.RandomState(100 + len(draws))
        draws.append(rs.standard_normal(shape).astype(np.float32))
        return draws[-1]

    worker = SVDWorker(denoise, fe_gpu, width=W, height=H, out_width=64, out_height=48, num_frames=T,
                       num_inference_steps=3, noise_fn=noise_fn)
    rs = np.random.RandomState(0)
    req = {"b_action": np.array([[4, 2, 1, 3], [4, 1, 3, 3]], dtype=np.int64),
           "save_dirs": [str(tmp_path / "a"), str(tmp_path / "b")], "request_model_name": "igen",
           "b_image": rs.randint(0, 256, size=(2, 3, H, W), dtype=np.uint8), "return_objects": [True, True]}
    out = worker(req)
    pf = out["pred_frames"]
    assert pf.shape == (2, T, 3, 48, 64) and pf.dtype == np.uint8

    # the same chain on the CPU: fp32 frontend + oracle loop, same draws
    fe_cpu = VO.TorchFrontend(vsd, clip, device="cpu", vae_dtype=torch.float32, dtype=torch.float32, **vcfg)
    _, _, _, images = P.parse_request(req)
    x = np.stack([P.preprocess_image(im, W, H) for im in images])
    il, ie = fe_cpu.encode(x, draws[0], 0.02)
    lat = O.denoise({k: torch.from_numpy(v) for k, v in sd.items()}, cfg.as_dict(), torch.from_numpy(il), torch.from_numpy(ie),
                    torch.from_numpy(draws[1]), req["b_action"], num_steps=3).numpy()
    frames = fe_cpu.decode(lat)
    ref = P.build_response(P.images_to_tensor([P.frames_to_pil(f) for f in frames], save_size=(64, 48)),
                           req["b_action"], req["save_dirs"], True)["pred_frames"]
    diff = np.abs(pf.astype(np.int32) - ref.astype(np.int32))
    mse = float((diff.astype(np.float64) ** 2).mean())
    psnr = 10 * np.log10(255.0 ** 2 / max(mse, 1e-12))
    print(f"[parity] end-to-end uint8 frames: mean|diff|={diff.mean():.3f} levels, max={diff.max()}, PSNR={psnr:.1f} dB")
    assert psnr > psnr_gate   # per-pixel metric of the reference (evaluation/FVD/calculate_psnr.py:6-15); FVD needs absent I3D weights


def test_serve_worker_full_size_over_tcp(tmp_path):
    """The launcher at production geometry (576x1024x14, full-width UNet / VAE / CLIP, random-init weights):
    one 2-candidate request through the TCP transport of solver_base.py:645-688, 2 Euler steps."""
    import socket
    import threading

    import serve_worker
    from wiw_amd.server import protocol as PR
    from wiw_amd.server.worker import serve_tcp

    args = serve_worker.arg_parser().parse_args(["--random_weights", "--num_inference_steps", "2", "--port", "1"])
    worker = serve_worker.build_worker(args)
    ready, stop = threading.Event(), threading.Event()
    th = threading.Thread(target=serve_tcp, kwargs=dict(worker=worker, port=0, ready=ready, stop=stop), daemon=True)
    th.start()
    assert ready.wait(10)
    rs = np.random.RandomState(0)
    req = {"b_action": np.array([[4] + [1] * 13, [4] + [2] * 6 + [1] * 7], dtype=np.int64),
           "save_dirs": [str(tmp_path / "a"), str(tmp_path / "b")], "request_model_name": "igen",
           "b_image": rs.randint(0, 256, size=(2, 3, 576, 1024), dtype=np.uint8), "return_objects": [True, True]}
    with socket.create_connection(("127.0.0.1", ready.port)) as c:
        PR.write_framed(c, req)
        out = PR.read_framed(c)
        PR.write_framed(c, PR.DONE)
    stop.set()
    th.join(5)
    pf = out["pred_frames"]
    assert pf.shape == (2, 14, 3, 480, 480) and pf.dtype == np.uint8
    assert np.isfinite(pf.astype(np.float32)).all() and pf.std() > 0
    assert list(out["save_dirs"]) == req["save_dirs"]


def test_coalesced_clients_get_their_alone_bytes_on_the_hip_path(tmp_path):
    """SURVEY.md §8f row 3 on the GPU: two clients' requests (2 + 3 candidates) coalesced into ONE HIP batch of 5; every
    client's uint8 frames equal what it gets when served alone — bit for bit (batch-independent kernels)."""
    import threading

    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection

    import wiw_amd  # noqa: F401
    from wiw_amd import frontend as FE
    from wiw_amd.config import UNetConfig
    from wiw_amd.pipeline import SVDDenoiser
    from wiw_amd.server.worker import Coalescer, SVDWorker
    from wiw_amd.unet import UNetHIP
    from wiw_amd.vae import HIPFrontend, VAEHIP
    from wiw_amd.weights import random_state_dict

    T, H, W = 4, 128, 256
    cfg = UNetConfig.tiny(T)
    unet = UNetHIP(cfg, random_state_dict(cfg, 21), "cuda:0")
    den = SVDDenoiser(unet)
    vcfg = dict(block_out_channels=(64, 128, 128, 128), layers_per_block=1)
    torch.manual_seed(0)
    clip = CLIPVisionModelWithProjection(CLIPVisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                                                          num_attention_heads=2, image_size=224, patch_size=32,
                                                          projection_dim=1024, hidden_act="gelu")).eval()
    # VAE and CLIP both on the HIP kernels: every stage of the request is batch-independent bit for bit
    fe = HIPFrontend(VAEHIP(FE.vae_random_state_dict(22, **vcfg), "cuda:0", hip=unet.hip, **vcfg), clip, clip="hip")

    def denoise(il, ie, nz, act, **kw):
        return den.denoise(torch.as_tensor(il), torch.as_tensor(ie), torch.as_tensor(nz), act, **kw).cpu().numpy()

    # noise as a function of the candidate (its save_dir), so that batching does not change anybody's draws
    calls = []

    def make(noise_seed_base):
        return SVDWorker(denoise, fe, width=W, height=H, out_width=64, out_height=48, num_frames=T, num_inference_steps=2,
                         noise_fn=lambda shape: np.zeros(shape, np.float32))

    worker = make(0)

    def recording(req):
        calls.append(len(req["save_dirs"]))
        return worker(req)

    rs = np.random.RandomState(3)
    reqs = []
    for c, b in enumerate((2, 3)):
        reqs.append({"b_action": rs.randint(1, 4, size=(b, T)).astype(np.int64), "request_model_name": "igen",
                     "save_dirs": [str(tmp_path / f"c{c}_{i}") for i in range(b)],
                     "b_image": rs.randint(0, 256, size=(b, 3, H, W), dtype=np.uint8), "return_objects": [True] * b})
    alone = [worker(r)["pred_frames"] for r in reqs]
    co = Coalescer(recording, max_candidates=8, max_wait_s=2.0)
    outs = [None, None]
    ts = [threading.Thread(target=lambda i=i: outs.__setitem__(i, co.submit(reqs[i]))) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    co.close()
    assert calls == [5], f"the two requests must be evaluated as one batch of 5 candidates, got {calls}"
    for i in range(2):
        assert outs[i]["save_dirs"] == reqs[i]["save_dirs"]
        assert np.array_equal(outs[i]["pred_frames"], alone[i]), f"client {i}: coalesced frames differ from the alone run"
