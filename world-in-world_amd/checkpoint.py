"""Checkpoints of the fine-tuning loop: save / prune / resume as `FTsvd/train_svd.py` does.

Layout (train_svd.py:586-626, 1032-1062 — `accelerator.save_state` with the save hook writing `unet/`):

    <output_dir>/checkpoint-<global_step>/
        unet/config.json                           the config `save_pretrained` writes (train_svd.py:586-600 save hook): with the
                                                   weights file below, what the reference's own loaders need —
                                                   `UNetSpatioTemporalConditionModel.from_pretrained(<dir>, subfolder="unet", ...)`
                                                   (train_svd.py load hook, eval_inference.py:115-131)
        unet/diffusion_pytorch_model.safetensors   fp32 master parameters under the reference's parameter names; also read by
                                                   this package's serving loader (`weights.load_safetensors`, `serve_worker.py
                                                   --unet_path`)
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
UNET_CONFIG_FILE = os.path.join("unet", "config.json")
EMA_FILE = os.path.join("unet_ema", "diffusion_pytorch_model.safetensors")
EMA_CONFIG_FILE = os.path.join("unet_ema", "config.json")
DIFFUSERS_VERSION = "0.31.0"      # the vendored fork's version string (FTsvd/diffusers-private), as it writes it


def unet_config_dict(cfg) -> dict:
    """The `unet/config.json` of `UNetSpatioTemporalConditionModel.save_pretrained` for a `config.UNetConfig`: the
    registered __init__ arguments (unet_spatio_temporal_condition.py:72-97).  The fork's runtime kwargs (action_strategy,
    task_type, action_input_channel) are NOT stored there — the reference passes them at load time
    (eval_inference.py:116-125).  Pinned to files the reference itself wrote: tests/golden/unet_config*.json."""
    n = len(cfg.block_out_channels)
    return {
        "_class_name": "UNetSpatioTemporalConditionModel",
        "_diffusers_version": DIFFUSERS_VERSION,
        "addition_time_embed_dim": cfg.addition_time_embed_dim,
        "block_out_channels": list(cfg.block_out_channels),
        "cross_attention_dim": cfg.cross_attention_dim,
        "down_block_types": ["CrossAttnDownBlockSpatioTemporal"] * (n - 1) + ["DownBlockSpatioTemporal"],
        "in_channels": cfg.in_channels,
        "layers_per_block": cfg.layers_per_block,
        "num_attention_heads": list(cfg.num_attention_heads),
        "num_frames": cfg.num_frames,
        "out_channels": cfg.out_channels,
        "projection_class_embeddings_input_dim": cfg.projection_class_embeddings_input_dim,
        "sample_size": None,
        "transformer_layers_per_block": 1,
        "up_block_types": ["UpBlockSpatioTemporal"] + ["CrossAttnUpBlockSpatioTemporal"] * (n - 1),
    }


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
                    total_limit: Optional[int] = None, unet_config: Optional[dict] = None, ema: Optional[dict] = None) -> str:
    """Write checkpoint-<global_step>.  master: the fp32 parameters (rank 0 writes them; pass None on other ranks);
    optimizer: flat name -> tensor dict (this rank's part when sharded); unet_config: `unet_config_dict(cfg)`, written as
    unet/config.json next to the weights (the file set of `save_pretrained`); ema: {"shadow": name -> tensor, "state": dict}
    of `train_unet.EMAShadow` (`--use_ema`).  Returns the directory."""
    from safetensors.torch import save_file

    if ema is not None and unet_config is None:    # BEFORE anything is written: a partial checkpoint-N would be resumed from
        raise ValueError("save_checkpoint: `ema` needs `unet_config` — unet_ema/config.json must be loadable as a model "
                         "directory (EMAModel.from_pretrained reads the architecture from it)")
    path = os.path.join(output_dir, f"checkpoint-{global_step}")
    os.makedirs(os.path.join(path, "unet"), exist_ok=True)
    if rank == 0 and master is not None:
        save_file({k: v.detach().to("cpu", torch.float32).contiguous() for k, v in master.items()}, os.path.join(path, UNET_FILE))
        if unet_config is not None:
            with open(os.path.join(path, UNET_CONFIG_FILE), "w") as f:       # same formatting as diffusers' to_json_string
                f.write(json.dumps(unet_config, indent=2, sort_keys=True) + "\n")
        with open(os.path.join(path, "trainer_state.json"), "w") as f:
            json.dump(dict(meta, global_step=int(global_step)), f, indent=1, sort_keys=True)
    if rank == 0 and ema is not None:
        # `ema_unet.save_pretrained(<dir>/unet_ema)` (train_svd.py:588-589): the averaged weights as a model directory whose
        # config.json carries the EMA state next to the architecture (training_utils.py:390-403 registers it into the config)
        os.makedirs(os.path.join(path, "unet_ema"), exist_ok=True)
        save_file({k: v.detach().to("cpu", torch.float32).contiguous() for k, v in ema["shadow"].items()},
                  os.path.join(path, EMA_FILE))
        with open(os.path.join(path, EMA_CONFIG_FILE), "w") as f:
            f.write(json.dumps(dict(unet_config or {}, **ema["state"]), indent=2, sort_keys=True) + "\n")
    name = f"optimizer_rank{rank}.safetensors" if sharded else "optimizer.safetensors"
    if rank == 0 or sharded:
        save_file({k: v.detach().to("cpu").contiguous() for k, v in optimizer.items()}, os.path.join(path, name))
    if rank == 0:
        prune(output_dir, total_limit)
    return path


EMA_STATE_KEYS = ("decay", "min_decay", "optimization_step", "update_after_step", "use_ema_warmup", "inv_gamma", "power")


def load_ema(path: str):
    """-> (shadow parameters, EMA state) of `checkpoint-<n>/unet_ema` (`EMAModel.from_pretrained`, train_svd.py:600-604)."""
    from safetensors.torch import load_file

    shadow = load_file(os.path.join(path, EMA_FILE))
    with open(os.path.join(path, EMA_CONFIG_FILE)) as f:
        cfg = json.load(f)
    return shadow, {k: cfg[k] for k in EMA_STATE_KEYS}


def load_checkpoint(path: str, rank: int = 0, sharded: bool = False) -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor], dict]:
    """-> (master parameters, optimizer tensors, trainer state).  Raises FileNotFoundError when a part is missing."""
    from safetensors.torch import load_file

    master = load_file(os.path.join(path, UNET_FILE))
    name = f"optimizer_rank{rank}.safetensors" if sharded else "optimizer.safetensors"
    optimizer = load_file(os.path.join(path, name))
    with open(os.path.join(path, "trainer_state.json")) as f:
        meta = json.load(f)
    return master, optimizer, meta
