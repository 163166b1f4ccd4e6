"""Candidate sharding across ranks (world_size 2, gloo, CPU): the N>1 path of bench.py / the server."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import wiw_amd  # noqa: F401
from wiw_amd.parallel import shard_bounds, sharded_denoise

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fake_denoise(il, ie, noise, actions, num_steps=2, **kw):
    a = torch.as_tensor(np.asarray(actions), dtype=torch.float64)
    if a.dim() == 3:                       # manipulation rows (B, T, 8): every value matters
        a = (a * torch.arange(1, 9, dtype=torch.float64)).sum(-1)
    a = a.to(torch.float32)[:, :, None, None, None]
    return il[:, None] * (1 + a) + 0.5 * noise + ie.mean((1, 2))[:, None, None, None, None] * num_steps


def _run(rank, world, port, B, q, manip=False):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        req = [None] * 4
        if rank == 0:
            g = torch.Generator().manual_seed(0)
            acts = np.arange(B * 3).reshape(B, 3) % 5
            if manip:     # (B, T, 8) continuous rows keep their shape and their float values through the scatter
                acts = np.random.RandomState(3).standard_normal((B, 3, 8))
            req = [torch.randn(B, 4, 4, 8, generator=g), torch.randn(B, 1, 16, generator=g),
                   torch.randn(B, 3, 4, 4, 8, generator=g), acts]
        out = sharded_denoise(fake_denoise, torch.device("cpu"), *req, num_steps=3)
        if rank == 0:
            ref = fake_denoise(req[0], req[1], req[2], req[3], num_steps=3)
            q.put(float((out - ref).abs().max()))
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_bounds():
    assert shard_bounds(64, 8) == [(8 * i, 8 * i + 8) for i in range(8)]
    assert shard_bounds(3, 2) == [(0, 2), (2, 3)]
    assert shard_bounds(1, 4) == [(0, 1), (1, 1), (1, 1), (1, 1)]


def _spawn(B, manip=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, 2, port, B, q, manip)) for r in range(2)]
    for p in procs:
        p.start()
    err = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return err


def test_sharded_equals_single_rank_even_and_ragged():
    assert _spawn(4) == 0.0   # 2 + 2
    assert _spawn(3) == 0.0   # 2 + 1 (ragged)
    assert _spawn(1) == 0.0   # 1 + 0 (a rank with no candidate)
    assert _spawn(3, manip=True) == 0.0   # manipulation: (B, T, 8) float actions (were cast to int64 (T,))


# ----------------------------------------------------------------------------------------------------------------
# ShardedWorker: request dict sharded over ranks, responses concatenated in candidate order
# ----------------------------------------------------------------------------------------------------------------
def fake_worker(req):
    a = np.asarray(req["b_action"])
    frames = (np.asarray(req["b_image"]).astype(np.int64).sum(axis=(1, 2, 3)) % 251)[:, None] + a
    return {"save_dirs": list(req["save_dirs"]), "pred_frames": frames.astype(np.uint8)}


def _run_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from wiw_amd.parallel import ShardedWorker
    try:
        sw = ShardedWorker(fake_worker)
        if rank == 0:
            errs = []
            for B in (5, 2, 1):
                rs = np.random.RandomState(B)
                req = {"b_action": rs.randint(0, 5, size=(B, 14)), "save_dirs": [f"/tmp/c{i}" for i in range(B)],
                       "request_model_name": "igen",
                       "b_image": rs.randint(0, 256, size=(B, 3, 8, 16), dtype=np.uint8), "return_objects": [True] * B}
                out, ref = sw(req), fake_worker(req)
                errs.append(out["save_dirs"] == ref["save_dirs"] and np.array_equal(out["pred_frames"], ref["pred_frames"]))
            sw.close()
            q.put(all(errs))
        else:
            sw.follow()
    finally:
        dist.destroy_process_group()


def _poisoned_worker(req):
    if any("poison" in d for d in req["save_dirs"]):
        raise ValueError("bad candidate")
    return fake_worker(req)


def _run_worker_errors(rank, world, port, q):
    """A slice that raises must not desynchronise the group: rank 0 gets ShardedWorkerError, the NEXT request works."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from wiw_amd.parallel import ShardedWorker, ShardedWorkerError
    from wiw_amd.server import plumbing as P
    try:
        sw = ShardedWorker(_poisoned_worker, validate=lambda r: P.validate_request(r, 14))
        if rank == 0:
            rs = np.random.RandomState(0)

            def mk(B, poison=None):
                dirs = [f"/tmp/c{i}" for i in range(B)]
                if poison is not None:
                    dirs[poison] = "/tmp/poison"
                return {"b_action": rs.randint(0, 5, size=(B, 14)), "save_dirs": dirs, "request_model_name": "igen",
                        "b_image": rs.randint(0, 256, size=(B, 3, 8, 16), dtype=np.uint8), "return_objects": [True] * B}

            res = []
            for poison in (3, 0):            # the failing candidate lands on rank 1, then on rank 0
                try:
                    sw(mk(4, poison))
                    res.append("no error")
                except ShardedWorkerError as e:
                    res.append("bad candidate" in str(e))
            # malformed request: rejected on rank 0 before any collective (b_action with 13 columns)
            bad = mk(2)
            bad["b_action"] = bad["b_action"][:, :13]
            try:
                sw(bad)
                res.append("no error")
            except AssertionError:
                res.append(True)
            good = mk(3)
            out, ref = sw(good), fake_worker(good)
            res.append(out["save_dirs"] == ref["save_dirs"] and np.array_equal(out["pred_frames"], ref["pred_frames"]))
            sw.close()
            q.put(res)
        else:
            sw.follow()
    finally:
        dist.destroy_process_group()


def test_sharded_worker_survives_a_failing_slice():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run_worker_errors, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    assert q.get(timeout=120) == [True, True, True, True]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0


def test_sharded_worker_matches_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    assert q.get(timeout=120) is True
    for p in procs:
        p.join(60)
        assert p.exitcode == 0


# ----------------------------------------------------------------------------------------------------------------
# ShardedAdamW (row f2): reduce-scatter of gradients, sharded AdamW, all-gather of parameters == AdamW on the mean gradient
# ----------------------------------------------------------------------------------------------------------------
def _adamw_torch(p, g, m, v, step, lr, b1, b2, eps, wd):      # the update Hip.adamw_step performs on the GPU
    p.mul_(1 - lr * wd)
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    p.addcdiv_(m / (1 - b1 ** step), (v / (1 - b2 ** step)).sqrt() + eps, value=-lr)


def _zero_run(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from wiw_amd.parallel import ShardedAdamW

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shapes = {"a.weight": (37, 5), "a.bias": (37,), "b.weight": (64, 33), "c": (1,)}       # ragged sizes, 3 buckets
        g0 = torch.Generator().manual_seed(0)
        init = {k: torch.randn(*s, generator=g0) for k, s in shapes.items()}
        opt = ShardedAdamW(shapes, torch.device("cpu"), _adamw_torch, bucket_elems=1024, lr=1e-2, weight_decay=0.05)
        opt.load(init)
        assert opt.n_buckets == 3 and opt.master.numel() == 3 * opt.slice
        ref = {k: v.clone().requires_grad_(True) for k, v in init.items()}
        ropt = torch.optim.AdamW(list(ref.values()), lr=1e-2, weight_decay=0.05)
        for step in range(4):
            mean_grads = {}
            for k, s in reversed(list(shapes.items())):               # a backward produces the last parameters first
                per_rank = [torch.randn(*s, generator=torch.Generator().manual_seed(100 * step + 10 * r + len(k))) for r in range(world)]
                if step >= 2:                                          # gradients handed over as they complete: a bucket goes
                    opt.notify(k, per_rank[rank].reshape(-1) if step == 3 else per_rank[rank], set(shapes))   # out asynchronously
                else:
                    opt.view(opt.grads, k).copy_(per_rank[rank])      # this rank's local gradient
                mean_grads[k] = sum(per_rank) / world
            if step == 1:
                opt.reduce_bucket(opt.n_buckets - 1)                   # a bucket handed over early, synchronously
            if step >= 2:
                assert opt.async_launched == (step - 1) * opt.n_buckets and all(opt._reduced)
            opt.step()
            for k in shapes:
                ref[k].grad = mean_grads[k]
            ropt.step()
        err = max(float((opt.view(opt.params, k) - ref[k].detach()).abs().max()) for k in shapes)
        # a gradient that was NOT announced and arrives after its bucket went out is an error, not a silent drop
        for k in ("b.weight", "a.bias", "a.weight"):
            opt.notify(k, torch.zeros(shapes[k]), set(shapes) - {"c"})
        try:
            opt.notify("c", torch.zeros(1), set(shapes) - {"c"})
            late = False
        except RuntimeError:
            late = True
        opt.step()
        assert late
        full = [None] * world
        dist.all_gather_object(full, opt.params.clone())
        if rank == 0:
            q.put((err, bool(all(torch.equal(full[0], f) for f in full))))
    finally:
        dist.destroy_process_group()


def test_sharded_adamw_equals_adamw_on_the_mean_gradient():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_zero_run, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    err, same = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert err <= 2e-6 and same       # every rank holds the same updated parameters
