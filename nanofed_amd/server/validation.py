"""Update validation and signing (reference: nanofed/server/validation.py:15-212).

``DefaultModelValidator`` screens client updates before aggregation:
shape check against reference shapes, range check (all values finite and every
tensor's L2 norm <= ``max_norm``) and a statistical check (z-score of the
update's global L2 norm against the norms of reference updates; skipped with
fewer than ``min_clients_for_stats`` references).

For arena-backed updates (``ArenaStateDict``) the range and statistics checks
read one fused finite-and-sum-of-squares reduction per update
(``nanofed_amd.ops.finite_and_sqnorm``) instead of looping over tensors; the
per-tensor norms needed by ``validate_range`` come from one segmented reduce
over the arena.

``SecurityManager`` signs updates with RSA-2048 / PSS-SHA256 over the
concatenation ``name:bytes`` of the name-sorted state (fp32, CPU).

``validation_mask`` turns verdicts into participation weights (0 for
rejected clients) for the device-plane reduce.
"""

from __future__ import annotations

from dataclasses import dataclass
from enum import Enum, auto
from typing import Protocol, Sequence

import torch

from nanofed_amd.core import ModelUpdate
from nanofed_amd.crypto import CryptoError, RSAKey
from nanofed_amd.utils import Logger


class ValidationResult(Enum):
    VALID = auto()
    INVALID_SHAPE = auto()
    INVALID_RANGE = auto()
    INVALID_SIGNATURE = auto()
    ANOMALOUS = auto()


@dataclass(frozen=True)
class ValidationConfig:
    max_norm: float = 10.0
    max_update_size: int = 1024 * 1024 * 100
    min_clients_for_stats: int = 5
    z_score_threshold: float = 2.0
    signature_required: bool = True


class ModelValidator(Protocol):
    def validate_shape(
        self, update: ModelUpdate, reference: dict[str, torch.Size]
    ) -> ValidationResult: ...
    def validate_range(
        self, update: ModelUpdate, config: ValidationConfig
    ) -> ValidationResult: ...
    def validate_statistics(
        self, update: ModelUpdate, reference_updates: Sequence[ModelUpdate]
    ) -> ValidationResult: ...
    def validate_signature(
        self, update: ModelUpdate, public_key: bytes
    ) -> ValidationResult: ...


def _global_norm(state: dict[str, torch.Tensor]) -> float:
    flat = getattr(state, "flat", None)
    if isinstance(flat, torch.Tensor):
        from nanofed_amd import ops

        sq, _ = ops.finite_and_sqnorm(flat)
        return float(sq) ** 0.5
    tensors = [t.detach().float() for t in state.values()]
    if not tensors:
        return 0.0
    return float(
        torch.linalg.vector_norm(torch.stack(list(torch._foreach_norm(tensors))))
    )


class DefaultModelValidator:
    def __init__(self, config: ValidationConfig) -> None:
        self._config = config
        self._logger = Logger()

    def validate_shape(
        self, update: ModelUpdate, reference: dict[str, torch.Size]
    ) -> ValidationResult:
        try:
            state = update["model_state"]
            for name, shape in reference.items():
                if name not in state:
                    self._logger.warning(f"Missing parameter: {name}")
                    return ValidationResult.INVALID_SHAPE
                got = state[name].shape
                if tuple(got) != tuple(shape):
                    self._logger.warning(
                        f"Shape mismatch for {name}: got {got}, expected {shape}"
                    )
                    return ValidationResult.INVALID_SHAPE
            return ValidationResult.VALID
        except Exception as exc:
            self._logger.error(f"Shape validation failed: {exc}")
            return ValidationResult.INVALID_SHAPE

    def validate_range(
        self, update: ModelUpdate, config: ValidationConfig
    ) -> ValidationResult:
        try:
            state = update["model_state"]
            flat = getattr(state, "flat", None)
            if isinstance(flat, torch.Tensor):
                from nanofed_amd import ops

                _, bad = ops.finite_and_sqnorm(flat)
                if int(bad) > 0:
                    return ValidationResult.INVALID_RANGE
                norms = state.per_tensor_norms()  # type: ignore[attr-defined]
                if bool((norms > config.max_norm).any()):
                    return ValidationResult.INVALID_RANGE
                return ValidationResult.VALID
            for tensor in state.values():
                t = tensor.detach().float()
                if not bool(torch.isfinite(t).all()):
                    return ValidationResult.INVALID_RANGE
                if float(torch.linalg.vector_norm(t)) > config.max_norm:
                    return ValidationResult.INVALID_RANGE
            return ValidationResult.VALID
        except Exception as exc:
            self._logger.error(f"Range validation failed: {exc}")
            return ValidationResult.INVALID_RANGE

    def validate_statistics(
        self, update: ModelUpdate, reference_updates: Sequence[ModelUpdate]
    ) -> ValidationResult:
        if len(reference_updates) < self._config.min_clients_for_stats:
            return ValidationResult.VALID
        try:
            ref = torch.tensor(
                [_global_norm(r["model_state"]) for r in reference_updates]
            )
            z = abs(_global_norm(update["model_state"]) - ref.mean()) / (
                ref.std() + 1e-8
            )
            if float(z) > self._config.z_score_threshold:
                return ValidationResult.ANOMALOUS
            return ValidationResult.VALID
        except Exception as exc:
            self._logger.error(f"Statistical validation failed: {exc}")
            return ValidationResult.ANOMALOUS

    def validate_all(
        self,
        update: ModelUpdate,
        reference_shapes: dict[str, torch.Size] | None = None,
        reference_updates: Sequence[ModelUpdate] = (),
    ) -> ValidationResult:
        """Shape, then range, then statistics; first failure wins."""
        if reference_shapes is not None:
            verdict = self.validate_shape(update, reference_shapes)
            if verdict is not ValidationResult.VALID:
                return verdict
        verdict = self.validate_range(update, self._config)
        if verdict is not ValidationResult.VALID:
            return verdict
        return self.validate_statistics(update, reference_updates)


def validation_mask(verdicts: Sequence[ValidationResult]) -> list[float]:
    """1.0 for accepted updates, 0.0 for rejected ones."""
    return [1.0 if v is ValidationResult.VALID else 0.0 for v in verdicts]


def _signing_message(update: ModelUpdate) -> bytes:
    parts = []
    for name, tensor in sorted(update["model_state"].items()):
        raw = tensor.detach().to("cpu").contiguous().numpy().tobytes()
        parts.append(name.encode("utf-8") + b":" + raw)
    return b"".join(parts)


class SecurityManager:
    def __init__(self) -> None:
        self._key = RSAKey.generate(2048)
        self._logger = Logger()

    def get_public_key(self) -> bytes:
        """PEM (SubjectPublicKeyInfo) for client-side verification."""
        return self._key.public_pem()

    def sign_update(self, update: ModelUpdate) -> bytes:
        try:
            return self._key.pss_sign(_signing_message(update))
        except Exception as exc:
            self._logger.error(f"Failed to sign update: {exc}")
            raise

    def verify_signature(
        self, update: ModelUpdate, signature: bytes, public_key: bytes
    ) -> bool:
        try:
            key = RSAKey.from_public_pem(public_key)
            return key.pss_verify(_signing_message(update), signature)
        except CryptoError as exc:
            self._logger.error(f"Signature verification failed: {exc}")
            return False
        except Exception as exc:
            self._logger.error(f"Signature verification failed: {exc}")
            return False
