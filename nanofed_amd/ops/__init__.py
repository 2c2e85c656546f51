"""Device operations (hand-written gfx950 HIP kernels behind torch-tensor
handles).  CPU tensors run fp32 PyTorch implementations of the same
arithmetic; GPU tensors always take the HIP path.
"""

from __future__ import annotations

import torch

from .arena_ops import (
    MODE_LAPLACE,
    MODE_NORMAL,
    MODE_SYM_UNIFORM,
    MODE_UNIFORM,
    clip_noise,
    l2_norm,
    PeerMaskPlan,
    pairwise_mask_,
    pairwise_repair_,
    philox_fill,
    round_weights,
    scale_accumulate,
    screen_rows,
    select_copy,
    sgd_multi,
    sgd_step,
    sqnorm_finite,
    weighted_sum,
)
from .library import hip_lib


def finite_and_sqnorm(flat: torch.Tensor) -> tuple[float, int]:
    """Host values ``(sum of squares of finite entries, non-finite count)``."""
    res = sqnorm_finite(flat.reshape(-1)).tolist()
    return float(res[0]), int(res[1])


def mask_add_(flat: torch.Tensor, seed: int, scale: float, offset: int = 0) -> torch.Tensor:
    """``flat += scale * U(-1, 1)`` drawn from Philox(seed, offset): the
    pairwise blinding mask of secure aggregation.  Two parties that share
    ``seed`` generate identical masks, so ``+scale`` on one and ``-scale`` on
    the other cancel exactly in the sum."""
    return philox_fill(
        flat, seed=seed, offset=offset, mode=MODE_SYM_UNIFORM, scale=scale,
        accumulate=True,
    )


__all__ = [
    "MODE_LAPLACE",
    "MODE_NORMAL",
    "MODE_SYM_UNIFORM",
    "MODE_UNIFORM",
    "clip_noise",
    "finite_and_sqnorm",
    "hip_lib",
    "l2_norm",
    "mask_add_",
    "PeerMaskPlan",
    "pairwise_mask_",
    "pairwise_repair_",
    "philox_fill",
    "round_weights",
    "scale_accumulate",
    "screen_rows",
    "select_copy",
    "sgd_multi",
    "sgd_step",
    "sqnorm_finite",
    "weighted_sum",
]
