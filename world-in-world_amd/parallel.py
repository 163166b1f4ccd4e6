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
