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


def _rank_main(rank, world, port, q):
    import sys

    import torch.distributed as dist

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import wiw_amd  # noqa: F401
    from wiw_amd.config import UNetConfig
    from wiw_amd.parallel import ShardedWorker, sharded_denoise
    from wiw_amd.pipeline import SVDDenoiser
    from wiw_amd.unet import UNetHIP
    from wiw_amd.weights import random_state_dict

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        cfg = UNetConfig.tiny(4)
        den = SVDDenoiser(UNetHIP(cfg, random_state_dict(cfg, 5), dev))
        B, T, h, w = 5, 4, 16, 32
        req = [None] * 4
        if rank == 0:
            rs = np.random.RandomState(0)
            req = [torch.from_numpy(rs.standard_normal((B, 4, h, w)).astype(np.float32)).to(dev),
                   torch.from_numpy(rs.standard_normal((B, 1, cfg.cross_attention_dim)).astype(np.float32)).to(dev),
                   torch.from_numpy(rs.standard_normal((B, T, 4, h, w)).astype(np.float32)).to(dev),
                   np.array([[4, 2, 1, 3], [4, 1, 1, 1], [4, 3, 3, 2], [4, 1, 2, 1], [4, 3, 1, 1]])]
        out = sharded_denoise(den.denoise, dev, *req, num_steps=2)

        def worker(r):
            return {"save_dirs": list(r["save_dirs"]), "pred_frames": np.asarray(r["b_image"])[:, :1] + 1}

        sw = ShardedWorker(worker)
        if rank == 0:
            ref = den.denoise(*req, num_steps=2)
            ok = torch.equal(out, ref.float())       # candidates are evaluated with batch-independent arithmetic
            rs = np.random.RandomState(1)
            wreq = {"b_action": req[3], "save_dirs": list("abcde"), "request_model_name": "igen",
                    "b_image": rs.randint(0, 200, size=(5, 3, 8, 16), dtype=np.uint8), "return_objects": [True] * 5}
            got = sw(wreq)
            ok = ok and got["save_dirs"] == list("abcde") and np.array_equal(got["pred_frames"], wreq["b_image"][:, :1] + 1)
            sw.close()
            q.put(bool(ok))
        else:
            sw.follow()
    finally:
        dist.destroy_process_group()


def test_sharded_paths_over_rccl_two_ranks():
    """Two processes, two GPUs, RCCL over xGMI: scatter / gather of candidate slices; skipped on a one-GPU box."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the driver's test box has one; the multi-rank logic runs under gloo on CPU)")
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    assert q.get(timeout=300) is True
    for p in procs:
        p.join(60)
        assert p.exitcode == 0


def _torchrun_bench(extra, timeout=600):
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 ... bench.py --gpus 1 ...` exactly as the driver launches
    the N > 1 runs (rendezvous on 127.0.0.1, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher), with
    WIW_FORCE_DIST=1 so that the ONE rank takes the multi-rank code path: env-driven `init_process_group("nccl")`, scatter /
    gather of candidate slices (`parallel.sharded_denoise`) or the ZeRO-1 optimiser (`parallel.ShardedAdamW`), barrier +
    max-over-ranks timing.  Returns the JSON line."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WIW_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", *extra]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    return json.loads(lines[0])


def test_bench_under_torchrun_takes_the_rccl_path_with_one_rank():
    """the multi-GPU entry of bench.py run the way the driver runs it, on the one GPU a test box has."""
    res = _torchrun_bench(["--tiny", "--height", "128", "--width", "256", "--num-inference-steps", "2", "--steps", "1",
                           "--warmup", "1", "--batch", "2", "--no-cpu-baseline", "--no-kernel-events"])
    assert res["n_gpus"] == 1 and res["value"] > 0 and "rccl process group, 1 rank" in res["config"]["collectives"]
    assert res["scaling"] == "weak" and res["config"]["candidates_per_gpu"] == 2
    # strong-scaling form: a fixed request of 3 candidates sharded over the (one) rank
    res = _torchrun_bench(["--tiny", "--height", "128", "--width", "256", "--num-inference-steps", "2", "--steps", "1",
                           "--warmup", "0", "--total-candidates", "3", "--no-cpu-baseline", "--no-kernel-events"])
    assert res["scaling"] == "strong" and res["value"] > 0


def test_train_bench_under_torchrun_uses_the_sharded_optimizer():
    res = _torchrun_bench(["--train", "--tiny", "--train-height", "256", "--train-width", "512", "--steps", "2", "--warmup", "1",
                           "--no-autotune"])
    assert res["n_gpus"] == 1 and res["value"] > 0 and "ZeRO-1" in res["config"]["parallelism"]
    assert np.isfinite(res["final_loss"])
