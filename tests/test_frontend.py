"""PyTorch frontend (VAE encode / temporal decode, CLIP preprocessing) against vectors produced by the
reference modules (oracle/make_golden.py: gen_frontend).  CPU, fp32."""
import numpy as np
import torch

import wiw_amd  # noqa: F401
import vae_oracle as VO
from wiw_amd import frontend as FE

CFG = dict(block_out_channels=(32, 64, 64, 64), layers_per_block=1)


def _rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / (np.abs(b).max() + 1e-30))


def test_vae_encode_mode_and_temporal_decode(golden):
    g = golden("frontend_tiny.npz")
    sd = {k: torch.from_numpy(v) for k, v in FE.vae_random_state_dict(int(g["weight_seed"]), **CFG).items()}
    mode = VO.vae_encode_mode(sd, torch.from_numpy(g["image"]), 4, 1)
    assert _rel(mode.numpy(), g["latent_mode"]) < 2e-5
    dec = VO.vae_decode(sd, torch.from_numpy(g["latents"]), int(g["num_frames"]), 4, 1)
    assert _rel(dec.numpy(), g["decoded"]) < 2e-5


def test_resize_with_antialiasing(golden):
    g = golden("frontend_tiny.npz")
    out = FE.resize_with_antialiasing(torch.from_numpy(g["resize_in"]), (224, 224))
    assert _rel(out.numpy(), g["resize_out"]) < 1e-5
    # the 576x1024 -> 224 path (real serving shape) is checked on a strided sample of the reference output;
    # replay the generator's draws (oracle/make_golden.py: gen_frontend) up to the 576x1024 image
    rs = np.random.RandomState(int(g["resize_big_seed"]))
    rs.uniform(-1, 1, size=(2, 3, 32, 64)); rs.standard_normal((8, 4, 4, 8)); rs.uniform(-1, 1, size=(1, 3, 72, 128))
    small = rs.uniform(-1, 1, size=(1, 3, 576, 1024)).astype(np.float32)
    out2 = FE.resize_with_antialiasing(torch.from_numpy(small), (224, 224)).numpy()[:, :, ::16, ::16]
    assert _rel(out2, g["resize_big_out_sample"]) < 1e-5


def test_clip_preprocess_range_and_spec():
    x = torch.zeros(1, 3, 64, 128)
    y = FE.clip_preprocess(x)   # 0 in [-1,1] -> 0.5 in [0,1] -> (0.5 - mean) / std
    exp = [(0.5 - m) / s for m, s in zip(FE.CLIP_MEAN, FE.CLIP_STD)]
    assert y.shape == (1, 3, 224, 224) and np.allclose(y[0, :, 100, 100].numpy(), exp, atol=1e-5)
    assert len(FE.vae_state_dict_spec()) == 248 or len(FE.vae_state_dict_spec()) > 200


def test_frontend_object_roundtrip_cpu(golden):
    """TorchFrontend plumbing on CPU tensors (device-agnostic torch code; the GPU run uses the same path)."""
    g = golden("frontend_tiny.npz")
    sd = FE.vae_random_state_dict(int(g["weight_seed"]), **CFG)

    class FakeClip(torch.nn.Module):
        def forward(self, pixel_values):
            class O:
                image_embeds = pixel_values.float().mean((2, 3)).repeat(1, 342)[:, :1024]
            return O()

    fe = VO.TorchFrontend(sd, FakeClip(), device="cpu", vae_dtype=torch.float32, dtype=torch.float32, **CFG)
    lat, emb = fe.encode(g["image"], np.zeros_like(g["image"]), 0.02)
    assert _rel(lat, g["latent_mode"]) < 2e-5 and emb.shape == (2, 1, 1024)
    fr = fe.decode(g["latents"].reshape(2, 4, 4, 4, 8) * fe.scaling_factor)
    assert _rel(fr.reshape(8, 3, 32, 64), g["decoded"]) < 2e-5
