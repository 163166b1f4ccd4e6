"""Candidate-sharded denoising across the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference scales by running N independent worker processes behind a manager that splits every
request into batch-1 tasks over pipes (downstream/utils/worker_manager.py:448-469, 555-570).  Here the
candidates of ONE request are sharded across ranks: rank 0 owns the request, broadcasts the (small)
conditioning tensors, every rank denoises its contiguous slice with a full weight replica, and the
latents are gathered back on rank 0.  There is no exchange inside the loop (candidates are
independent); the two collectives move ~1 MB per candidate.

`backend="nccl"` is RCCL on ROCm; tests run the same code with `gloo` on CPU tensors.
"""
from __future__ import annotations

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


def _bcast(t: Optional[torch.Tensor], shape, dtype, device, src=0) -> torch.Tensor:
    if t is None:
        t = torch.empty(shape, dtype=dtype, device=device)
    else:
        t = t.to(device=device, dtype=dtype).contiguous()
    dist.broadcast(t, src=src)
    return t


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
        meta[:6] = torch.tensor([B, T, h, w, image_embeddings.shape[-1], image_embeddings.shape[-2]])
    dist.broadcast(meta, src=0)
    B, T, h, w, D, L = (int(v) for v in meta[:6].tolist())
    il = _bcast(image_latents, (B, 4, h, w), torch.float32, device)
    ie = _bcast(image_embeddings, (B, L, D), torch.float32, device)
    nz = _bcast(noise, (B, T, 4, h, w), torch.float32, device)
    ac = _bcast(None if actions is None else torch.as_tensor(np.asarray(actions), dtype=torch.int64), (B, T),
                torch.int64, device)
    bounds = shard_bounds(B, world)
    lo, hi = bounds[rank]
    if hi > lo:
        mine = denoise_fn(il[lo:hi], ie[lo:hi], nz[lo:hi], ac[lo:hi].cpu().numpy(), **kw).to(torch.float32).contiguous()
    else:
        mine = torch.empty((0, T, 4, h, w), dtype=torch.float32, device=device)
    # ragged gather as a padded all_gather (available on both RCCL and gloo)
    width = max(b - a for a, b in bounds)
    pad = torch.zeros((width, T, 4, h, w), dtype=torch.float32, device=device)
    pad[: hi - lo] = mine
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    if rank != 0:
        return None
    return torch.cat([parts[k][: b - a] for k, (a, b) in enumerate(bounds)])


# ------------------------------------------------------------------------------------------------
# Serving: one request sharded over the ranks, every stage (encode, denoise, decode, PIL post-processing) local
# ------------------------------------------------------------------------------------------------
class ShardedWorker:
    """Rank 0 owns the client connection; a request's candidates are sliced over the ranks exactly as the reference
    manager slices them over worker processes (every value `v[lo:hi]`, worker_manager.py:448-469), each rank runs the
    WHOLE worker on its slice — VAE encode, denoise, VAE decode and the CPU-side PIL post-processing all scale with
    the rank count ("decode-sharded") — and the per-rank response dicts come back to rank 0, where
    they are concatenated in candidate order.  The two object collectives move the request (1.8 MB / candidate) and
    the uint8 frames (9.7 MB / candidate); nothing is exchanged inside the loop.

        rank 0:   ShardedWorker(worker)(request) -> response            (e.g. behind server.worker.serve_tcp)
        rank > 0: ShardedWorker(worker).follow()                        (returns when rank 0 calls close())
    """

    _STOP = "__wiw_stop__"

    def __init__(self, worker: Callable[[dict], dict]):
        self.worker = worker
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    @staticmethod
    def _slice(req: dict, lo: int, hi: int) -> dict:
        # strings (request_model_name) stay whole: the reference manager slices them too ("igen"[i:i+1]), which its
        # worker only survives because the name check is dead code there
        return {k: (v if isinstance(v, str) else v[lo:hi]) for k, v in req.items()}

    @staticmethod
    def _concat(parts: List[dict]) -> dict:
        out: dict = {}
        for part in parts:
            for k, v in part.items():
                out.setdefault(k, []).append(v)
        res = {}
        for k, vs in out.items():
            res[k] = np.concatenate(vs) if isinstance(vs[0], np.ndarray) else [x for v in vs for x in v]
        return res

    def _step(self, request) -> Optional[dict]:
        box = [request]
        dist.broadcast_object_list(box, src=0)
        req = box[0]
        if isinstance(req, str) and req == self._STOP:
            return None
        n = len(next(v for v in req.values() if not isinstance(v, str)))
        lo, hi = shard_bounds(n, self.world)[self.rank]
        mine = self.worker(self._slice(req, lo, hi)) if hi > lo else {}
        gathered = [None] * self.world if self.rank == 0 else None
        dist.gather_object(mine, gathered, dst=0)
        if self.rank != 0:
            return {}
        return self._concat([g for g in gathered if g])

    def __call__(self, request: dict) -> dict:
        assert self.rank == 0, "only rank 0 takes client requests"
        return self._step(request)

    def follow(self) -> None:
        assert self.rank != 0
        while self._step(None) is not None:
            pass

    def close(self) -> None:
        if self.rank == 0:
            dist.broadcast_object_list([self._STOP], src=0)
