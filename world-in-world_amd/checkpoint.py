"""Checkpoints of the fine-tuning loop: save / prune / resume as `FTsvd/train_svd.py` does.

Layout (train_svd.py:586-626, 1032-1062 — `accelerator.save_state` with the save hook writing `unet/`):

    <output_dir>/checkpoint-<global_step>/
        unet/diffusion_pytorch_model.safetensors   fp32 master parameters under the reference's parameter names: the file
                                                   `UNetSpatioTemporalConditionModel.from_pretrained(<dir>/unet)` and this
                                                   package's serving loader (`weights.load_safetensors`, `serve_worker.py
                                                   --unet_path`) read
        optimizer.safetensors                      AdamW moments `exp_avg.<name>` / `exp_avg_sq.<name>` (single process), or
        optimizer_rank<r>.safetensors              this rank's ZeRO-1 slices of the flat buffers (`parallel.ShardedAdamW`)
        trainer_state.json                         global_step, micro-batch counter, loss scale, world size

"latest" resolves to the directory with the largest step (train_svd.py:795-799); after a save at most
`checkpoints_total_limit` directories are kept, oldest removed first (:1040-1062).  Host-side files only: nothing here
touches the GPU beyond copying tensors to the host.
"""
from __future__ import annotations

import json
import os
import shutil
from typing import Dict, Optional, Tuple

import torch

UNET_FILE = os.path.join("unet", "diffusion_pytorch_model.safetensors")


def _step_of(name: str) -> int:
    return int(name.split("-")[1])


def list_checkpoints(output_dir: str):
    """Checkpoint directory names under output_dir, oldest first (train_svd.py:796-798)."""
    if not os.path.isdir(output_dir):
        return []
    return sorted((d for d in os.listdir(output_dir) if d.startswith("checkpoint")), key=_step_of)


def resolve_resume(output_dir: str, resume_from_checkpoint: Optional[str]) -> Optional[str]:
    """The directory `--resume_from_checkpoint` names: None / "None" -> no resume; "latest" -> the newest checkpoint of
    output_dir (None if there is none: a new run starts, train_svd.py:801-805); anything else -> its basename under
    output_dir (:793)."""
    if not resume_from_checkpoint or resume_from_checkpoint == "None":
        return None
    if resume_from_checkpoint == "latest":
        dirs = list_checkpoints(output_dir)
        return os.path.join(output_dir, dirs[-1]) if dirs else None
    path = os.path.join(output_dir, os.path.basename(resume_from_checkpoint.rstrip("/")))
    return path if os.path.isdir(path) else None


def prune(output_dir: str, total_limit: Optional[int]):
    """Keep at most total_limit checkpoints, removing the oldest (train_svd.py:1040-1062).  Returns the removed names."""
    if total_limit is None:
        return []
    dirs = list_checkpoints(output_dir)
    removed = dirs[: max(0, len(dirs) - total_limit)]
    for d in removed:
        shutil.rmtree(os.path.join(output_dir, d))
    return removed


def save_checkpoint(output_dir: str, global_step: int, master: Optional[Dict[str, torch.Tensor]],
                    optimizer: Dict[str, torch.Tensor], meta: dict, rank: int = 0, sharded: bool = False,
                    total_limit: Optional[int] = None) -> str:
    """Write checkpoint-<global_step>.  master: the fp32 parameters (rank 0 writes them; pass None on other ranks);
    optimizer: flat name -> tensor dict (this rank's part when sharded).  Returns the directory."""
    from safetensors.torch import save_file

    path = os.path.join(output_dir, f"checkpoint-{global_step}")
    os.makedirs(os.path.join(path, "unet"), exist_ok=True)
    if rank == 0 and master is not None:
        save_file({k: v.detach().to("cpu", torch.float32).contiguous() for k, v in master.items()}, os.path.join(path, UNET_FILE))
        with open(os.path.join(path, "trainer_state.json"), "w") as f:
            json.dump(dict(meta, global_step=int(global_step)), f, indent=1, sort_keys=True)
    name = f"optimizer_rank{rank}.safetensors" if sharded else "optimizer.safetensors"
    if rank == 0 or sharded:
        save_file({k: v.detach().to("cpu").contiguous() for k, v in optimizer.items()}, os.path.join(path, name))
    if rank == 0:
        prune(output_dir, total_limit)
    return path


def load_checkpoint(path: str, rank: int = 0, sharded: bool = False) -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor], dict]:
    """-> (master parameters, optimizer tensors, trainer state).  Raises FileNotFoundError when a part is missing."""
    from safetensors.torch import load_file

    master = load_file(os.path.join(path, UNET_FILE))
    name = f"optimizer_rank{rank}.safetensors" if sharded else "optimizer.safetensors"
    optimizer = load_file(os.path.join(path, name))
    with open(os.path.join(path, "trainer_state.json")) as f:
        meta = json.load(f)
    return master, optimizer, meta
