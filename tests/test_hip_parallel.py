"""The RCCL code path of parallel.sharded_denoise / ShardedWorker on a real device: a one-rank `nccl` process group
(the driver's GPU box has one GPU; the multi-rank logic is covered by the world_size-2 gloo tests).  What this pins is
that every collective is issued on device tensors RCCL accepts and that the sharded result is the unsharded one, bit
for bit."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_sharded_paths_over_rccl_single_rank():
    import torch.distributed as dist

    import wiw_amd  # noqa: F401
    from wiw_amd.config import UNetConfig
    from wiw_amd.parallel import ShardedWorker, sharded_denoise
    from wiw_amd.pipeline import SVDDenoiser
    from wiw_amd.unet import UNetHIP
    from wiw_amd.weights import random_state_dict

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        cfg = UNetConfig.tiny(4)
        den = SVDDenoiser(UNetHIP(cfg, random_state_dict(cfg, 5), dev))
        rs = np.random.RandomState(0)
        B, T, h, w = 3, 4, 16, 32
        il = torch.from_numpy(rs.standard_normal((B, 4, h, w)).astype(np.float32)).to(dev)
        ie = torch.from_numpy(rs.standard_normal((B, 1, cfg.cross_attention_dim)).astype(np.float32)).to(dev)
        nz = torch.from_numpy(rs.standard_normal((B, T, 4, h, w)).astype(np.float32)).to(dev)
        acts = np.array([[4, 2, 1, 3], [4, 1, 1, 1], [4, 3, 3, 2]])
        out = sharded_denoise(den.denoise, dev, il, ie, nz, acts, num_steps=2)
        ref = den.denoise(il, ie, nz, acts, num_steps=2)
        assert torch.equal(out, ref.float())

        def worker(req):   # request-level sharding: object collectives over RCCL
            return {"save_dirs": list(req["save_dirs"]), "pred_frames": np.asarray(req["b_image"])[:, :1] + 1}

        sw = ShardedWorker(worker)
        req = {"b_action": acts, "save_dirs": ["a", "b", "c"], "request_model_name": "igen",
               "b_image": rs.randint(0, 200, size=(3, 3, 8, 16), dtype=np.uint8), "return_objects": [True] * 3}
        got = sw(req)
        assert got["save_dirs"] == ["a", "b", "c"] and np.array_equal(got["pred_frames"], req["b_image"][:, :1] + 1)
        sw.close()
    finally:
        dist.destroy_process_group()
