"""`train_unet.Trainer` + `parallel.ShardedAdamW` at world size 2 (gloo, CPU) around a STUB network: what is exercised is the
distributed glue that a one-GPU box cannot run — gradient hand-over during the backward (asynchronous bucket reductions
launched in the same order on every rank), gradient accumulation with local (no_sync) micro-batches, the sharded update
against `torch.optim.AdamW` on the mean gradient, and the GLOBAL overflow skip of the fp16 path (one rank's non-finite
gradient must make every rank skip, or the collectives of the optimiser step deadlock).  The network is three tensors with
a quadratic loss whose gradients are known in closed form; the HIP kernels play no part (`TrainStep` is stubbed)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _adamw_torch(p, g, m, v, step, lr, b1, b2, eps, wd):
    p.mul_(1 - lr * wd)
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    p.addcdiv_(m / (1 - b1 ** step), (v / (1 - b2 ** step)).sqrt() + eps, value=-lr)


class _Tape:
    after_op = None


class _StubNet:
    """loss = 0.5 * sum_k c_k |w_k - t|^2 with a per-sample target t: dL/dw_k = c_k (w_k - t).  `backward` produces the gradients
    LAST parameter first, calling the tape hook after each — as the real tape does after every closure."""
    dt, device = torch.bfloat16, torch.device("cpu")

    def __init__(self, poison_rank=None):
        g = torch.Generator().manual_seed(3)
        self.master = {"a.weight": torch.randn(40, 7, generator=g), "b.bias": torch.randn(33, generator=g), "c.weight": torch.randn(64, 9, generator=g),
                       "frozen.weight": torch.randn(5, 5, generator=g)}
        self.coef = {"a.weight": 1.0, "b.bias": 0.5, "c.weight": 2.0}
        self.tape, self.grads, self.wants, self.hip, self._direct = _Tape(), {}, (lambda n: True), None, {}
        self.refreshed, self.poison = 0, poison_rank

    def forward(self, target, *_):
        self.grads, self._t = {}, float(target)
        return torch.zeros(1)

    def backward(self, dpred, loss_scale=1.0):
        for k in reversed(list(self.coef)):
            g = self.coef[k] * (self.master[k] - self._t)
            if self.poison is not None and dist.get_rank() == self.poison and k == "b.bias":
                g = g * float("inf")
            self.grads[k] = g
            if self.tape.after_op is not None:
                self.tape.after_op()
        return self.grads

    def refresh(self, names=None):
        self.refreshed += 1


class _St:
    def __init__(self, target):
        self.unet_input, self.timestep, self.ehs, self.added_time_ids, self.action_ids = target, None, None, None, None


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    import conftest  # noqa: F401  (registers the package alias)
    import wiw_amd  # noqa: F401
    from wiw_amd import train as T
    from wiw_amd.parallel import ShardedAdamW
    from wiw_amd.train_unet import Trainer

    class _TrainStep:                      # the EDM loss kernel is not part of this test
        def __init__(self, hip):
            pass

        def loss_and_grad(self, pred, st):
            return torch.tensor(1.0), torch.zeros(1)
    T.TrainStep = _TrainStep
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lr, accum = 1e-2, 2
        net = _StubNet()
        shapes = {k: tuple(v.shape) for k, v in net.master.items()}
        opt = ShardedAdamW(shapes, torch.device("cpu"), _adamw_torch, bucket_elems=256, lr=lr, weight_decay=0.05)
        assert opt.n_buckets >= 4
        tr = Trainer(net, lr=lr, weight_decay=0.05, optimizer=opt, grad_accum=accum, loss_scale=1.0)
        ref = {k: v.clone().requires_grad_(True) for k, v in net.master.items() if k in net.coef}
        ropt = torch.optim.AdamW(list(ref.values()), lr=lr, weight_decay=0.05)
        frozen0 = net.master["frozen.weight"].clone()
        micro = 0
        for window in range(3):
            mean = {k: torch.zeros_like(v) for k, v in ref.items()}
            for mb in range(accum):
                for r in range(world):     # every rank's target of this micro-batch (this rank uses its own)
                    t = 0.1 * (1 + r) + 0.01 * micro
                    for k in ref:
                        mean[k] += net.coef[k] * (ref[k].detach() - t) / (world * accum)
                before = opt.async_launched
                tr.step(_St(0.1 * (1 + rank) + 0.01 * micro))
                micro += 1
                if mb < accum - 1:
                    assert opt.async_launched == before and tr.steps == window       # local accumulation: no collective, no update
            assert tr.steps == window + 1
            if window >= 1:                # the first window learns which parameters get a gradient; later ones overlap
                assert opt.async_launched > 0
            for k in ref:
                ref[k].grad = mean[k]
            ropt.step()
        err = max(float((net.master[k] - ref[k].detach()).abs().max()) for k in ref)
        assert torch.equal(net.master["frozen.weight"], frozen0)                       # no gradient: never read back
        # ---- fp16-style overflow on ONE rank: both ranks skip the update and halve the scale (no deadlock)
        net2 = _StubNet(poison_rank=1)
        opt2 = ShardedAdamW(shapes, torch.device("cpu"), _adamw_torch, bucket_elems=256, lr=lr)
        tr2 = Trainer(net2, lr=lr, optimizer=opt2, loss_scale=4.0)
        start = {k: v.clone() for k, v in net2.master.items()}
        tr2.step(_St(0.3))
        skipped = tr2.steps == 0 and tr2.loss_scale == 2.0 and all(torch.equal(net2.master[k], start[k]) for k in start)
        net2.poison = None
        tr2.step(_St(0.3))
        moved = tr2.steps == 1 and not torch.equal(net2.master["a.weight"], start["a.weight"])
        same = [None] * world
        dist.all_gather_object(same, {k: v.clone() for k, v in net.master.items()})
        if rank == 0:
            q.put((err, skipped, moved, all(torch.equal(same[0][k], same[1][k]) for k in same[0])))
        elif not (skipped and moved):
            q.put((err, skipped, moved, False))
    finally:
        dist.destroy_process_group()


def test_trainer_glue_world_size_2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    err, skipped, moved, same = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert err <= 5e-6, err                 # == AdamW on the mean gradient over ranks and micro-batches
    assert skipped and moved and same       # global overflow skip; every rank holds the same parameters


def test_sharded_adamw_bucket_layout_properties():
    """Property test (hypothesis) of the flat bucket layout on a one-rank gloo group in this process: for random parameter
    shapes, bucket sizes and hand-over orders (synchronous copy, `notify` in any order, a subset of parameters without a
    gradient), two steps of `ShardedAdamW` equal `torch.optim.AdamW` on the same gradients; a parameter that got no gradient
    is updated as AdamW updates one with a ZERO gradient (the flat update cannot skip it — `Trainer` does not read those back)."""
    from hypothesis import given, settings
    from hypothesis import strategies as st

    sys.path.insert(0, ROOT)
    import wiw_amd  # noqa: F401
    from wiw_amd.parallel import ShardedAdamW

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        shape = st.lists(st.integers(1, 9), min_size=1, max_size=3).map(tuple)

        @settings(max_examples=40, deadline=None)
        @given(st.lists(shape, min_size=1, max_size=7), st.integers(4, 300), st.randoms(use_true_random=False), st.booleans())
        def run(shapes, bucket, rnd, use_notify):
            names = [f"p{i}" for i in range(len(shapes))]
            spec = dict(zip(names, shapes))
            g0 = torch.Generator().manual_seed(len(shapes) * 1000 + bucket)
            init = {k: torch.randn(*s, generator=g0) for k, s in spec.items()}
            opt = ShardedAdamW(spec, torch.device("cpu"), _adamw_torch, bucket_elems=bucket, lr=1e-2, weight_decay=0.03)
            opt.load(init)
            ref = {k: v.clone().requires_grad_(True) for k, v in init.items()}
            ropt = torch.optim.AdamW(list(ref.values()), lr=1e-2, weight_decay=0.03)
            live = [k for k in names if rnd.random() < 0.8] or names[:1]
            for step in range(2):
                order = list(live)
                rnd.shuffle(order)
                for k in order:
                    g = torch.randn(*spec[k], generator=g0)
                    if use_notify:
                        opt.notify(k, g, set(live))
                    else:
                        opt.view(opt.grads, k).copy_(g)
                    ref[k].grad = g
                for k in names:
                    if k not in live:
                        ref[k].grad = torch.zeros(spec[k])
                opt.step()
                ropt.step()
            for k in names:
                assert float((opt.view(opt.params, k) - ref[k].detach()).abs().max()) <= 2e-6, (k, shapes, bucket)
        run()
    finally:
        dist.destroy_process_group()


def test_dead_parameter_rule_matches_the_reference_autograd():
    """`Trainer.is_dead` (which parameters an optimiser must NOT hold: torch.optim.AdamW never touches a parameter without a
    gradient, a flat optimiser over everything would weight-decay it) against the gradient norms of the reference's own
    `loss.backward()` (tests/golden/train_step_tiny.npz): dead <=> the reference gradient is (round-off) zero."""
    import numpy as np

    import wiw_amd  # noqa: F401
    from wiw_amd.train_unet import Trainer

    g = np.load(os.path.join(ROOT, "tests", "golden", "train_step_tiny.npz"), allow_pickle=True)
    names, norms = [str(n) for n in g["grad_names"]], g["grad_norms"]
    med = float(np.median(norms[norms > 0]))
    dead = [n for n in names if Trainer.is_dead(n)]
    assert len(dead) == 132
    for n, nr in zip(names, norms):      # dead: autograd round-off only (<= 1e-9); live: a gradient (a scalar mix_factor may be 1e-8)
        assert (nr <= 5e-9) if Trainer.is_dead(n) else (nr >= 1e-8), (n, nr, med)      # (median live norm: 5.7e-4)

    class _Net:
        master = {n: torch.zeros(1) for n in names}
    full = Trainer.optimizer_shapes(_Net, "full")
    assert set(full) == set(names) - set(dead) and list(full) == [n for n in names if n in full]      # model order kept
    new = Trainer.optimizer_shapes(_Net, "new")
    assert new and all(("action" in n) or ("noise" in n) for n in new)


def _subset_worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    import conftest  # noqa: F401  (registers the package alias)
    import wiw_amd  # noqa: F401
    from wiw_amd import train as T
    from wiw_amd.parallel import ShardedAdamW
    from wiw_amd.train_unet import Trainer

    class _TrainStep:                      # the EDM loss kernel is not part of this test
        def __init__(self, hip):
            pass

        def loss_and_grad(self, pred, st):
            return torch.tensor(1.0), torch.zeros(1)
    T.TrainStep = _TrainStep

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        net = _StubNet()
        shapes = {k: tuple(v.shape) for k, v in net.master.items() if k != "frozen.weight"}     # the optimiser holds a SUBSET
        opt = ShardedAdamW(shapes, torch.device("cpu"), _adamw_torch, bucket_elems=256, lr=1e-2, weight_decay=0.5)
        tr = Trainer(net, lr=1e-2, weight_decay=0.5, optimizer=opt, loss_scale=4.0, scale_growth_interval=2)
        frozen0 = net.master["frozen.weight"].clone()
        assert "frozen.weight" not in opt.offsets and not tr.trainable("frozen.weight") and tr.trainable("a.weight")
        scales = []
        for i in range(5):
            tr.step(_St(0.1 * (i + rank)))
            scales.append(tr.loss_scale)
        ok = torch.equal(net.master["frozen.weight"], frozen0) and scales == [4.0, 8.0, 8.0, 16.0, 16.0]    # x2 every 2 good steps
        # layout fingerprint: same construction -> same hash; another bucket size -> another hash (resume would refuse it)
        same = ShardedAdamW(shapes, torch.device("cpu"), _adamw_torch, bucket_elems=256).layout_fingerprint() == opt.layout_fingerprint()
        other = ShardedAdamW(shapes, torch.device("cpu"), _adamw_torch, bucket_elems=512).layout_fingerprint() != opt.layout_fingerprint()
        import tempfile

        d = tempfile.mkdtemp(prefix=f"ck{rank}_") if rank else None
        box = [d]
        dist.broadcast_object_list(box, src=1 if world > 1 else 0)
        out = box[0] or tempfile.mkdtemp()
        path = tr.save(out)
        dist.barrier()
        tr.load(path)                                                   # same layout: accepted
        net3 = _StubNet()
        tr3 = Trainer(net3, lr=1e-2, optimizer=ShardedAdamW(shapes, torch.device("cpu"), _adamw_torch, bucket_elems=512), loss_scale=4.0)
        refused = False
        try:
            tr3.load(path)
        except AssertionError as e:
            refused = "flat layout" in str(e)
        q.put((rank, ok, same, other, refused, tr.loss_scale))
    finally:
        dist.destroy_process_group()


def test_optimizer_subset_loss_scale_growth_and_layout_fingerprint():
    """(i) a sharded optimiser built from the trainable / live shapes only — the frozen parameter is neither
    decayed nor read back; (ii) GradScaler-style growth of the fp16 loss scale (x2 after N good steps in a row); (iii) a
    resume under another flat layout is refused instead of silently permuting optimiser slices."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_subset_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, same, other, refused, scale in res:
        assert ok and same and other and refused and scale == 16.0, (rank, ok, same, other, refused, scale)
