"""Static configuration of the served model.

Mirrors the class defaults of `UNetSpatioTemporalConditionModel.__init__`
(reference: FTsvd/diffusers-private/diffusers/models/unets/unet_spatio_temporal_condition.py:72-97)
plus the fork's runtime kwargs passed at load time (FTsvd/eval_inference.py:116-125), and the
EulerDiscreteScheduler config of the stock SVD snapshot.
"""
from __future__ import annotations

from dataclasses import asdict, dataclass
from typing import Tuple


@dataclass(frozen=True)
class UNetConfig:
    in_channels: int = 8
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    num_attention_heads: Tuple[int, ...] = (5, 10, 20, 20)
    layers_per_block: int = 2
    cross_attention_dim: int = 1024
    addition_time_embed_dim: int = 256
    projection_class_embeddings_input_dim: int = 768
    num_frames: int = 14
    action_input_channel: int = 14  # == num_frames for task_type='navigation'
    action_strategy: str = "micro_cond"
    task_type: str = "navigation"

    def __post_init__(self):
        assert self.action_strategy == "micro_cond", "only the served strategy is implemented"
        assert len(self.block_out_channels) == len(self.num_attention_heads)
        for c, h in zip(self.block_out_channels, self.num_attention_heads):
            assert c % 32 == 0, "GroupNorm(32) needs C % 32 == 0"
            assert c // h == 64 and c % h == 0, "kernels are specialised for head_dim 64 (C/heads)"

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * 4

    def as_dict(self) -> dict:
        return asdict(self)

    @staticmethod
    def tiny(num_frames: int = 4) -> "UNetConfig":
        """Reduced-width config used by parity fixtures (same topology, head_dim 64)."""
        return UNetConfig(block_out_channels=(64, 128, 128, 128), num_attention_heads=(1, 2, 2, 2),
                          num_frames=num_frames, action_input_channel=num_frames)


@dataclass(frozen=True)
class SchedulerConfig:
    """EulerDiscreteScheduler(v_prediction, continuous timesteps, Karras sigmas) — SURVEY.md §8c."""
    sigma_min: float = 0.002
    sigma_max: float = 700.0
    rho: float = 7.0
