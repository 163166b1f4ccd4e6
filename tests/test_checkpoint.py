"""Checkpoint directory logic of the fine-tuning loop (row f2): the reference's layout, "latest" resolution and pruning
(`FTsvd/train_svd.py:791-808, 1032-1062`).  CPU only."""
import os

import torch

import wiw_amd  # noqa: F401
from wiw_amd import checkpoint as C


def _state(seed):
    g = torch.Generator().manual_seed(seed)
    master = {"conv_in.weight": torch.randn(8, 8, 3, 3, generator=g), "conv_in.bias": torch.randn(8, generator=g)}
    optim = {f"exp_avg.{k}": torch.randn_like(v) for k, v in master.items()}
    optim.update({f"exp_avg_sq.{k}": torch.rand_like(v) for k, v in master.items()})
    return master, optim


def test_round_trip_uses_the_reference_layout(tmp_path):
    master, optim = _state(0)
    path = C.save_checkpoint(str(tmp_path), 500, master, optim, {"micro": 2000, "loss_scale": 1.0, "world": 1})
    assert os.path.basename(path) == "checkpoint-500"
    assert os.path.isfile(os.path.join(path, "unet", "diffusion_pytorch_model.safetensors"))    # what from_pretrained(<dir>/unet) reads
    m2, o2, meta = C.load_checkpoint(path)
    assert meta["global_step"] == 500 and meta["micro"] == 2000
    assert all(torch.equal(master[k], m2[k]) for k in master) and all(torch.equal(optim[k], o2[k]) for k in optim)
    # the serving loader reads the same file
    from wiw_amd.weights import load_safetensors

    sd = load_safetensors(os.path.join(path, C.UNET_FILE))
    assert all(torch.equal(torch.as_tensor(sd[k]), master[k]) for k in master)


def test_latest_and_pruning(tmp_path):
    out = str(tmp_path)
    assert C.resolve_resume(out, "latest") is None and C.resolve_resume(out, None) is None and C.resolve_resume(out, "None") is None
    for step in (1000, 500, 2000):                 # numeric, not lexicographic, order: 500 < 1000 < 2000
        C.save_checkpoint(out, step, *_state(step), {"micro": 0, "loss_scale": 1.0, "world": 1})
    assert C.list_checkpoints(out) == ["checkpoint-500", "checkpoint-1000", "checkpoint-2000"]
    assert C.resolve_resume(out, "latest") == os.path.join(out, "checkpoint-2000")
    assert C.resolve_resume(out, "/somewhere/else/checkpoint-1000") == os.path.join(out, "checkpoint-1000")   # basename (:793)
    assert C.resolve_resume(out, "checkpoint-7") is None
    C.save_checkpoint(out, 3000, *_state(3), {"micro": 0, "loss_scale": 1.0, "world": 1}, total_limit=2)
    assert C.list_checkpoints(out) == ["checkpoint-2000", "checkpoint-3000"]          # oldest removed first
    C.save_checkpoint(out, 4000, *_state(4), {"micro": 0, "loss_scale": 1.0, "world": 1}, total_limit=1)   # train_svd.sh: limit 1
    assert C.list_checkpoints(out) == ["checkpoint-4000"]


def test_sharded_optimizer_files_are_per_rank(tmp_path):
    out = str(tmp_path)
    master, _ = _state(1)
    for rank in (0, 1):
        C.save_checkpoint(out, 10, master if rank == 0 else None, {"master": torch.full((4,), float(rank))},
                          {"micro": 0, "loss_scale": 1.0, "world": 2}, rank=rank, sharded=True)
    for rank in (0, 1):
        _, o, meta = C.load_checkpoint(os.path.join(out, "checkpoint-10"), rank=rank, sharded=True)
        assert meta["world"] == 2 and float(o["master"][0]) == float(rank)


def test_unet_config_json_is_the_file_the_reference_writes(tmp_path):
    """`unet/config.json` next to the weights, byte for byte what the reference's own `save_pretrained` / `save_config` writes
    (tests/golden/unet_config*.json: produced by running the reference class, oracle/make_golden.py `schema`), so its
    `from_pretrained(<dir>, subfolder="unet", ...)` (train_svd.py load hook, eval_inference.py:115-131) finds both files."""
    from wiw_amd.config import UNetConfig

    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for fname, cfg in (("unet_config.json", UNetConfig()), ("unet_config_tiny.json", UNetConfig.tiny(4))):
        master, optim = _state(5)
        path = C.save_checkpoint(str(tmp_path / fname), 7, master, optim, {"micro": 0, "loss_scale": 1.0, "world": 1},
                                 unet_config=C.unet_config_dict(cfg))
        assert sorted(os.listdir(os.path.join(path, "unet"))) == ["config.json", "diffusion_pytorch_model.safetensors"]
        with open(os.path.join(path, C.UNET_CONFIG_FILE)) as f, open(os.path.join(golden, fname)) as g:
            assert f.read() == g.read()
