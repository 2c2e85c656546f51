"""Immutable, range-checked privacy configuration
(reference: nanofed/privacy/config.py:17-85)."""

from __future__ import annotations

from enum import Enum, auto

from pydantic import BaseModel, ConfigDict, Field, field_validator

from . import constants as C


class NoiseType(Enum):
    GAUSSIAN = auto()
    LAPLACIAN = auto()


class PrivacyConfig(BaseModel):
    """(epsilon, delta) budget plus the clip norm and noise multiplier.

    Frozen: assigning to a field raises ``pydantic.ValidationError``.
    """

    model_config = ConfigDict(frozen=True)

    epsilon: float = Field(
        default=C.DEFAULT_EPSILON, ge=C.MIN_EPSILON, le=C.MAX_EPSILON,
        description="Privacy parameter epsilon",
    )
    delta: float = Field(
        default=C.DEFAULT_DELTA, ge=C.MIN_DELTA, le=C.MAX_DELTA,
        description="Privacy parameter delta",
    )
    max_gradient_norm: float = Field(
        default=C.DEFAULT_MAX_GRAD_NORM, gt=0,
        description="L2 clipping bound",
    )
    noise_multiplier: float = Field(
        default=C.DEFAULT_NOISE_MULTIPLIER, gt=0,
        description="Noise standard deviation in units of the clipping bound",
    )
    noise_type: NoiseType = Field(
        default=NoiseType.GAUSSIAN, description="Noise distribution"
    )

    @field_validator("epsilon")
    @classmethod
    def validate_epsilon(cls, v: float) -> float:
        if not (C.MIN_EPSILON <= v <= C.MAX_EPSILON):
            raise ValueError(
                f"epsilon must be between {C.MIN_EPSILON} and {C.MAX_EPSILON}"
            )
        return v

    @field_validator("delta")
    @classmethod
    def validate_delta(cls, v: float) -> float:
        if not (C.MIN_DELTA <= v <= C.MAX_DELTA):
            raise ValueError(
                f"delta must be between {C.MIN_DELTA} and {C.MAX_DELTA}"
            )
        return v
