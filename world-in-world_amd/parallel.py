"""Candidate-sharded denoising across the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference scales by running N independent worker processes behind a manager that splits every
request into batch-1 tasks over pipes (downstream/utils/worker_manager.py:448-469, 555-570).  Here the
candidates of ONE request are sharded across ranks: rank 0 owns the request, SCATTERS each rank's slice of the
(small) conditioning tensors, every rank denoises its contiguous slice with a full weight replica, and the
results are GATHERED on rank 0.  There is no exchange inside the loop (candidates are independent,
SURVEY.md §8e); the two collectives move ~1 MB per candidate (latents) or 9.7 MB per candidate (uint8 frames).

xGMI is point-to-point (7 links per GPU): scatter / gather from rank 0 use rank 0's 7 links in parallel, one slice
per link — nothing is relayed through a ring, and no rank receives bytes it does not need (a whole-request
broadcast would push N times the data over the same links).

`backend="nccl"` is RCCL on ROCm; tests run the same code with `gloo` on CPU tensors.
"""
from __future__ import annotations

import traceback
from typing import Callable, List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n_items: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, balanced partition: the first (n % world) ranks get one extra candidate."""
    q, r = divmod(n_items, world)
    out, lo = [], 0
    for k in range(world):
        hi = lo + q + (1 if k < r else 0)
        out.append((lo, hi))
        lo = hi
    return out


def pad_slices(t: torch.Tensor, bounds, width: int) -> List[torch.Tensor]:
    """Rows [lo, hi) of `t` for every rank, each zero-padded to `width` rows (collectives need equal shapes)."""
    parts = []
    for lo, hi in bounds:
        p = torch.zeros((width, *t.shape[1:]), dtype=t.dtype, device=t.device)
        p[: hi - lo] = t[lo:hi]
        parts.append(p)
    return parts


def unpad_concat(parts: List[torch.Tensor], bounds) -> torch.Tensor:
    """Inverse of `pad_slices`: candidate order is rank order."""
    return torch.cat([parts[k][: b - a] for k, (a, b) in enumerate(bounds)])


def _scatter_slices(t: Optional[torch.Tensor], item_shape, dtype, device, bounds, width: int) -> torch.Tensor:
    """Rank 0 holds `t` (B, *item_shape); every rank receives its rows [lo, hi) in a (width, *item_shape) buffer
    (zero-padded: scatter needs equal shapes)."""
    rank = dist.get_rank()
    mine = torch.empty((width, *item_shape), dtype=dtype, device=device)
    parts = None
    if rank == 0:
        parts = pad_slices(t.to(device=device, dtype=dtype), bounds, width)
    dist.scatter(mine, parts, src=0)
    return mine


def _gather_slices(mine: torch.Tensor, bounds, width: int) -> Optional[torch.Tensor]:
    """Inverse of `_scatter_slices`: rank r contributes rows [lo_r, hi_r); rank 0 returns their concatenation."""
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = bounds[rank]
    pad = torch.zeros((width, *mine.shape[1:]), dtype=mine.dtype, device=mine.device)
    pad[: hi - lo] = mine
    parts = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, parts, dst=0)
    if rank != 0:
        return None
    return unpad_concat(parts, bounds)


def sharded_denoise(denoise_fn: Callable[..., torch.Tensor], device: torch.device, image_latents: Optional[torch.Tensor],
                    image_embeddings: Optional[torch.Tensor], noise: Optional[torch.Tensor], actions: Optional[np.ndarray],
                    **kw) -> Optional[torch.Tensor]:
    """Rank 0 passes the full request (B candidates); other ranks pass None.  Returns the (B,T,4,h,w)
    latents on rank 0 (None elsewhere).  `denoise_fn(image_latents, image_embeddings, noise, actions, **kw)`
    is `SVDDenoiser.denoise` (or any stand-in with the same signature, used by the CPU tests)."""
    rank, world = dist.get_rank(), dist.get_world_size()
    meta = torch.zeros(8, dtype=torch.int64, device=device)
    if rank == 0:
        B, T, _, h, w = noise.shape
        meta[:7] = torch.tensor([B, T, h, w, image_embeddings.shape[-1], image_embeddings.shape[-2],
                                 1 if np.asarray(actions).ndim == 3 else 0])
    dist.broadcast(meta, src=0)
    B, T, h, w, D, L = (int(v) for v in meta[:6].tolist())
    bounds = shard_bounds(B, world)
    width = max(b - a for a, b in bounds)
    lo, hi = bounds[rank]
    n = hi - lo
    il = _scatter_slices(image_latents, (4, h, w), torch.float32, device, bounds, width)[:n]
    ie = _scatter_slices(image_embeddings, (L, D), torch.float32, device, bounds, width)[:n]
    nz = _scatter_slices(noise, (T, 4, h, w), torch.float32, device, bounds, width)[:n]
    # actions keep their own trailing shape and dtype: (B, T) int64 navigation ids, (B, T, 8) float manipulation rows — rank 0
    # announces which (the rank of the array selects the encoding, eval_inference.py:324-331)
    manip = bool(int(meta[6]))
    a_shape, a_dtype = ((T, 8), torch.float64) if manip else ((T,), torch.int64)
    ac = _scatter_slices(None if actions is None else torch.as_tensor(np.asarray(actions), dtype=a_dtype), a_shape,
                         a_dtype, device, bounds, width)[:n]
    if n > 0:
        mine = denoise_fn(il, ie, nz, ac.cpu().numpy(), **kw).to(torch.float32).contiguous()
    else:
        mine = torch.empty((0, T, 4, h, w), dtype=torch.float32, device=device)
    return _gather_slices(mine, bounds, width)


# ------------------------------------------------------------------------------------------------
# Serving: one request sharded over the ranks, every stage (encode, denoise, decode, PIL post-processing) local
# ------------------------------------------------------------------------------------------------
class ShardedWorkerError(RuntimeError):
    """A rank failed on its slice of a request; the process group stayed in step (every rank reached the gather)."""


class ShardedWorker:
    """Rank 0 owns the client connection; a request's candidates are sliced over the ranks exactly as the reference
    manager slices them over worker processes (every value `v[lo:hi]`, worker_manager.py:448-469), each rank runs the
    WHOLE worker on its slice — VAE encode, denoise, VAE decode and the CPU-side PIL post-processing all scale with
    the rank count ("decode-sharded") — and the responses come back to rank 0 in candidate order.

    Data movement per request: each rank receives ONLY its slice (`scatter_object_list`: 1.8 MB / candidate of uint8
    panorama); the uint8 frames come back as ONE padded tensor gather (9.7 MB / candidate, no pickling of the bulk
    data; on RCCL the gather runs device to device over xGMI); the small rest (save_dirs, status) is a gathered object.

    Failure containment (the reference lets a failing worker process die and the client hang, worker_manager.py:369-379):
    rank 0 validates the request BEFORE any collective; every rank catches exceptions of its slice and still takes
    part in both gathers, so one malformed request cannot desynchronise or kill the group — rank 0 raises
    `ShardedWorkerError` for that request and serves the next one.

        rank 0:   ShardedWorker(worker)(request) -> response            (e.g. behind server.worker.serve_tcp)
        rank > 0: ShardedWorker(worker).follow()                        (returns when rank 0 calls close())
    """

    _STOP = "__wiw_stop__"

    def __init__(self, worker: Callable[[dict], dict], device: Optional[torch.device] = None,
                 validate: Optional[Callable[[dict], None]] = None):
        self.worker = worker
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        # tensors of a collective live where the backend wants them: the GPU for RCCL, the host for gloo
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        self.device = device
        self.validate = validate

    @staticmethod
    def _slice(req: dict, lo: int, hi: int) -> dict:
        # strings (request_model_name) stay whole: the reference manager slices them too ("igen"[i:i+1]), which its
        # worker only survives because the name check is dead code there
        return {k: (v if isinstance(v, str) else v[lo:hi]) for k, v in req.items()}

    def _step(self, request) -> Optional[dict]:
        # ---- scatter: rank k receives (n_candidates, its slice of the request) — or the stop token
        if self.rank == 0:
            if isinstance(request, str):
                objs = [request] * self.world
            else:
                n = len(next(v for v in request.values() if not isinstance(v, str)))
                bounds = shard_bounds(n, self.world)
                objs = [(n, self._slice(request, lo, hi)) for lo, hi in bounds]
        else:
            objs = None
        box = [None]
        dist.scatter_object_list(box, objs, src=0)
        if isinstance(box[0], str) and box[0] == self._STOP:
            return None
        n, sub = box[0]
        bounds = shard_bounds(n, self.world)
        lo, hi = bounds[self.rank]
        width = max(b - a for a, b in bounds)
        # ---- run the slice; an exception becomes a status object, never a missed collective
        mine, err = {}, None
        if hi > lo:
            try:
                mine = self.worker(sub)
            except BaseException as e:  # noqa: BLE001 — the group must stay in step whatever the slice raised
                err = f"rank {self.rank}, candidates [{lo}, {hi}): {type(e).__name__}: {e}\n{traceback.format_exc(limit=6)}"
                mine = {}
        frames = mine.get("pred_frames") if isinstance(mine, dict) else None
        small = {k: v for k, v in mine.items() if k != "pred_frames"} if isinstance(mine, dict) else {}
        status = dict(err=err, small=small, frame_shape=None if frames is None else tuple(np.asarray(frames).shape[1:]))
        gathered = [None] * self.world if self.rank == 0 else None
        dist.gather_object(status, gathered, dst=0)
        # ---- every rank learns whether (and in which per-candidate shape) frames travel: one small broadcast
        plan = [None]
        if self.rank == 0:
            shapes = {g["frame_shape"] for g in gathered if g["frame_shape"] is not None}
            ok = all(g["err"] is None for g in gathered) and len(shapes) == 1 and \
                all(g["frame_shape"] is not None for g, (a, b) in zip(gathered, bounds) if b > a)
            plan = [shapes.pop() if ok else None]
        dist.broadcast_object_list(plan, src=0)
        out_frames = None
        if plan[0] is not None:
            t = torch.zeros((hi - lo, *plan[0]), dtype=torch.uint8, device=self.device)
            if hi > lo:
                t.copy_(torch.as_tensor(np.ascontiguousarray(frames)))
            out_frames = _gather_slices(t, bounds, width)
        if self.rank != 0:
            return {}
        errs = [g["err"] for g in gathered if g["err"]]
        if errs:
            raise ShardedWorkerError("request failed on %d rank(s):\n%s" % (len(errs), "\n".join(errs)))
        res: dict = {}
        for g, (a, b) in zip(gathered, bounds):
            if b > a:
                for k, v in g["small"].items():
                    res.setdefault(k, []).extend(v)
        if out_frames is not None:
            res["pred_frames"] = out_frames.cpu().numpy()
        return res

    def __call__(self, request: dict) -> dict:
        assert self.rank == 0, "only rank 0 takes client requests"
        if self.validate is not None:
            self.validate(request)      # raises BEFORE any collective: followers never see a malformed request
        return self._step(request)

    def follow(self) -> None:
        assert self.rank != 0
        while self._step(None) is not None:
            pass

    def close(self) -> None:
        if self.rank == 0:
            self._step(self._STOP)


# ------------------------------------------------------------------------------------------------
# Fine-tuning (row f2): data-parallel gradient reduction with the optimiser state sharded over the ranks (ZeRO stage 1, what the
# reference runs through accelerate + DeepSpeed: FTsvd/config/accelerate_deepspeed_o1_config.yaml, train_svd.py:954-968)
# ------------------------------------------------------------------------------------------------
class ShardedAdamW:
    """fp32 master weights, AdamW moments and the update are SHARDED over the ranks; gradients are averaged with one
    reduce-scatter per bucket and the updated parameters return with one all-gather per bucket:

        grads (fp32, flat, every rank)  --reduce_scatter(AVG) per bucket-->  rank r owns slice r of every bucket
        AdamW on the owned slices (update_fn: `Hip.adamw_step` on the GPU; a torch restatement in the CPU tests)
        params (flat, every rank)       <--all_gather per bucket--            the updated slices

    xGMI is point to point (7 links per GPU): reduce-scatter + all-gather move 2 (N-1)/N of the 6.1 GB fp32 gradient /
    3.06 GB 16-bit parameter stream over all links at once, where a ring all-reduce would be bound by one link; buckets
    (default 256 MiB: few, large collectives) are what a backward pass hands over as soon as their gradients are complete —
    `reduce_bucket(i)` can be called from the backward's stream order; `step()` reduces whatever is still pending.
    gloo (CPU tests) has no reduce-scatter: there the bucket is all-reduced and sliced, which is arithmetically the same.
    """

    def __init__(self, shapes, device, update_fn: Callable, bucket_elems: int = 64 * 1024 * 1024, lr: float = 1e-5,
                 betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device, self.update_fn = device, update_fn
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.offsets, off = {}, 0
        for name, shp in shapes.items():                      # flat layout in the given (model) order
            n = int(np.prod(shp))
            self.offsets[name] = (off, n, tuple(shp))
            off += n
        align = self.world * 4                                # every bucket splits into `world` 16-byte aligned slices
        self.bucket = max(align, (min(bucket_elems, off) + align - 1) // align * align)
        self.total = (off + self.bucket - 1) // self.bucket * self.bucket
        self.n_buckets = self.total // self.bucket
        self.slice = self.bucket // self.world
        self.grads = torch.zeros(self.total, dtype=torch.float32, device=device)          # the backward writes here
        self.params = torch.zeros(self.total, dtype=torch.float32, device=device)         # full copy (all-gathered)
        own = self.n_buckets * self.slice
        self.master = torch.zeros(own, dtype=torch.float32, device=device)                # owned slices, bucket-major
        self.m, self.v = torch.zeros_like(self.master), torch.zeros_like(self.master)
        self.gshard = torch.zeros_like(self.master)
        self.steps = 0
        self._reduced = [False] * self.n_buckets
        self._work = []                                       # in-flight asynchronous bucket reductions
        self.async_launched = 0                               # buckets handed over during a backward (tests / logs)
        # which parameters touch which bucket: a bucket is handed to the reduce-scatter as soon as ALL of them have a gradient
        self._bucket_names = [set() for _ in range(self.n_buckets)]
        for name, (o, n, _) in self.offsets.items():
            for b in range(o // self.bucket, (o + max(n, 1) - 1) // self.bucket + 1):
                self._bucket_names[b].add(name)
        self._pending = [set(x) for x in self._bucket_names]

    def notify(self, name: str, grad: torch.Tensor, expected: Optional[set] = None) -> None:
        """Called by the backward pass as soon as `name`'s gradient is complete: it is copied into the flat buffer, and every
        bucket whose parameters are now all in is reduce-scattered ASYNCHRONOUSLY while the backward goes on (the collective
        runs on RCCL's own stream; `step` waits for the handles).  `expected`: the names that will get a gradient this step
        (dead / frozen parameters are not waited for)."""
        dst = self.view(self.grads, name)
        dst.copy_(grad.reshape(dst.shape))
        o, n, _ = self.offsets[name]
        for b in range(o // self.bucket, (o + max(n, 1) - 1) // self.bucket + 1):
            if self._reduced[b]:
                raise RuntimeError(f"{name}: gradient arrived after its bucket {b} had been reduced (not in `expected`)")
            self._pending[b].discard(name)
            left = self._pending[b] if expected is None else (self._pending[b] & expected)
            if not left and not self._reduced[b]:
                self.reduce_bucket(b, async_op=True)

    def view(self, flat: torch.Tensor, name: str) -> torch.Tensor:
        off, n, shp = self.offsets[name]
        return flat[off:off + n].view(shp)

    def layout_fingerprint(self) -> str:
        """Hash of everything the flat layout depends on — the (name, offset, size) table in order, the bucket and total
        lengths and the world size.  Stored with a checkpoint and compared on resume: per-rank optimiser slices written
        under another bucket size or parameter order would otherwise be copied into permuted slots without any error."""
        import hashlib
        import json

        table = [[name, int(o), int(n)] for name, (o, n, _) in self.offsets.items()]
        blob = json.dumps({"table": table, "bucket": int(self.bucket), "total": int(self.total), "world": int(self.world)})
        return hashlib.sha1(blob.encode()).hexdigest()

    def load(self, state: dict) -> None:
        """Initial parameters (same on every rank) -> full copy + this rank's master slices.  `state` may hold more names
        than this optimiser owns (frozen / dead parameters are simply not part of it)."""
        for name in self.offsets:
            self.view(self.params, name).copy_(torch.as_tensor(state[name], dtype=torch.float32))
        for b in range(self.n_buckets):
            lo = b * self.bucket + self.rank * self.slice
            self.master[b * self.slice:(b + 1) * self.slice] = self.params[lo:lo + self.slice]

    def reduce_bucket(self, b: int, async_op: bool = False) -> None:
        """Average bucket b's gradients over the ranks; this rank keeps its slice."""
        g = self.grads[b * self.bucket:(b + 1) * self.bucket]
        out = self.gshard[b * self.slice:(b + 1) * self.slice]
        self._reduced[b] = True
        self.async_launched += int(async_op)
        if dist.get_backend() == "nccl":
            w = dist.reduce_scatter_tensor(out, g, op=dist.ReduceOp.AVG, async_op=async_op)
            if async_op:
                self._work.append((w, None))
        else:
            w = dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=async_op)
            fin = lambda: out.copy_(g[self.rank * self.slice:(self.rank + 1) * self.slice] / self.world)  # noqa: E731
            if async_op:
                self._work.append((w, fin))
            else:
                fin()

    def step(self) -> None:
        for w, fin in self._work:                             # buckets handed over during the backward
            w.wait()
            if fin is not None:
                fin()
        self._work = []
        for b in range(self.n_buckets - 1, -1, -1):           # the last parameters' gradients are complete first
            if not self._reduced[b]:
                self.reduce_bucket(b)
        self.steps += 1
        self.update_fn(self.master, self.gshard, self.m, self.v, self.steps, self.lr, self.betas[0], self.betas[1], self.eps, self.wd)
        for b in range(self.n_buckets):
            dist.all_gather_into_tensor(self.params[b * self.bucket:(b + 1) * self.bucket],
                                        self.master[b * self.slice:(b + 1) * self.slice])
        self.grads.zero_()
        self._reduced = [False] * self.n_buckets
        self._pending = [set(x) for x in self._bucket_names]
