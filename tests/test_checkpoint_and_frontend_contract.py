"""CPU tests of the loader / launcher contract (VERDICT r1 missing #8):
  * `serve_worker.resolve_unet_weights` finds the UNet where the reference's launch line puts it
    (`from_pretrained(unet_path, subfolder='unet')`, eval_inference.py:115-131; `--unet_path=.../checkpoint-6000`,
    workers_cfg.py:26) and refuses a bad path instead of silently serving the stock model;
  * the served UNet's 1 438-tensor state-dict schema == the reference class's (tests/golden/unet_schema.json, generated from
    the reference on the meta device by oracle/make_golden.py);
  * safetensors round trip of a (reduced-width) diffusers-style checkpoint in fp16 and bf16 through `load_safetensors`
    + `validate_state_dict`;
  * the worker hands CLIP the image at its ORIGINAL size (reference `_encode_image` runs on the un-resized PIL image,
    pipeline:192-199) and resizes only the VAE branch (pipeline:521).
"""
import json
import os

import numpy as np
import pytest
import torch

import wiw_amd  # noqa: F401
from wiw_amd.config import UNetConfig
from wiw_amd.server.worker import SVDWorker
from wiw_amd.weights import load_safetensors, random_state_dict, state_dict_spec, validate_state_dict

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_state_dict_schema_matches_reference_class():
    with open(os.path.join(GOLDEN, "unet_schema.json")) as f:
        ref = json.load(f)
    spec = state_dict_spec(UNetConfig())
    assert ref["n_tensors"] == len(ref["tensors"]) == 1438 and ref["n_params"] == 1528603626
    assert set(spec) == set(ref["tensors"]), sorted(set(spec) ^ set(ref["tensors"]))[:8]   # a safetensors file is unordered
    for k, shp in spec.items():
        assert list(shp) == ref["tensors"][k], k
    assert sum(int(np.prod(s)) for s in spec.values()) == ref["n_params"]


@pytest.mark.parametrize("dtype,fname", [(torch.float16, "diffusion_pytorch_model.fp16.safetensors"),
                                         (torch.bfloat16, "diffusion_pytorch_model.safetensors")])
def test_safetensors_round_trip_and_launcher_path_resolution(tmp_path, dtype, fname):
    from safetensors.torch import save_file

    import serve_worker

    cfg = UNetConfig.tiny(4)
    sd = {k: torch.from_numpy(v).to(dtype) for k, v in random_state_dict(cfg, 3).items()}
    ckpt = tmp_path / "checkpoint-6000"
    (ckpt / "unet").mkdir(parents=True)
    save_file(sd, str(ckpt / "unet" / fname))
    svd = tmp_path / "svd"
    (svd / "unet").mkdir(parents=True)
    save_file({k: v * 0 for k, v in sd.items()}, str(svd / "unet" / fname))
    # the reference launch line: --unet_path=<checkpoint dir holding unet/>
    path = serve_worker.resolve_unet_weights(str(ckpt), str(svd))
    assert path == str(ckpt / "unet" / fname)
    # a path that already is the unet folder
    assert serve_worker.resolve_unet_weights(str(ckpt / "unet"), str(svd)) == path
    # no --unet_path: the stock model of the snapshot
    assert serve_worker.resolve_unet_weights("", str(svd)) == str(svd / "unet" / fname)
    # a bad path is an error, never a silent fallback
    with pytest.raises(FileNotFoundError):
        serve_worker.resolve_unet_weights(str(tmp_path / "nope"), str(svd))
    (tmp_path / "empty").mkdir()
    with pytest.raises(FileNotFoundError):
        serve_worker.resolve_unet_weights(str(tmp_path / "empty"), str(svd))
    loaded = load_safetensors(path)
    validate_state_dict(cfg, loaded)
    assert all(loaded[k].dtype == dtype and torch.equal(loaded[k], sd[k]) for k in sd)
    bad = dict(loaded)
    bad["conv_in.weight"] = bad["conv_in.weight"][:, :4]
    with pytest.raises(ValueError):
        validate_state_dict(cfg, bad)
    del bad["conv_in.weight"]
    with pytest.raises(KeyError):
        validate_state_dict(cfg, bad)


class RecordingFrontend:
    def __init__(self):
        self.calls = []

    def encode(self, images, image_noise, noise_aug_strength, clip_images=None):
        self.calls.append((images.shape, None if clip_images is None else [c.shape for c in clip_images],
                           None if clip_images is None else [c.copy() for c in clip_images]))
        B, _, H, W = images.shape
        return np.zeros((B, 4, H // 8, W // 8), np.float32), np.zeros((B, 1, 1024), np.float32)

    def decode(self, latents):
        B, T, _, h, w = latents.shape
        return np.zeros((B, T, 3, 8 * h, 8 * w), np.float32)


def test_clip_branch_sees_the_original_size_image(tmp_path):
    fe = RecordingFrontend()
    worker = SVDWorker(lambda il, ie, nz, a, **kw: np.zeros_like(nz), fe, width=128, height=64, out_width=16, out_height=16)
    acts = np.tile(np.array([[4] + [1] * 13], dtype=np.int64), (2, 1))
    base = {"b_action": acts, "save_dirs": [str(tmp_path / "a"), str(tmp_path / "b")], "request_model_name": "igen",
            "return_objects": [True, True]}
    rs = np.random.RandomState(0)
    # (a) the image already has the model size: one tensor serves both branches
    worker(dict(base, b_image=rs.randint(0, 256, size=(2, 3, 64, 128), dtype=np.uint8)))
    assert fe.calls[-1][0] == (2, 3, 64, 128) and fe.calls[-1][1] is None
    # (b) a larger panorama: VAE branch LANCZOS-resized to 64x128, CLIP branch at 96x200 untouched
    img = rs.randint(0, 256, size=(2, 3, 96, 200), dtype=np.uint8)
    worker(dict(base, b_image=img))
    shape, clip_shapes, clip_arrays = fe.calls[-1]
    assert shape == (2, 3, 64, 128) and clip_shapes == [(3, 96, 200), (3, 96, 200)]
    assert np.array_equal(clip_arrays[1], 2.0 * (img[1].astype(np.float32) / 255.0) - 1.0)


def test_clip_preprocess_depends_on_the_input_size():
    """Resizing first changes what CLIP sees (the defect ADVICE r1 flagged): pin that the two routes differ, so the
    worker's choice matters, and that the un-resized route equals the reference fixture's resize."""
    from PIL import Image

    from wiw_amd import frontend as FE
    from wiw_amd.server import plumbing as P

    rs = np.random.RandomState(1)
    im = Image.fromarray(rs.randint(0, 256, size=(300, 500, 3), dtype=np.uint8))
    a = FE.clip_preprocess(torch.from_numpy(P.image_to_array(im))[None])
    b = FE.clip_preprocess(torch.from_numpy(P.preprocess_image(im, 256, 128))[None])
    assert a.shape == b.shape == (1, 3, 224, 224)
    assert float((a - b).abs().mean()) > 1e-2
