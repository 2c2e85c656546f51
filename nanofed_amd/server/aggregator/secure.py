"""Secure aggregation (reference: nanofed/server/aggregator/secure.py:18-313).

Three mechanisms:

``HomomorphicSecureAggregator``
    RSA-2048 / OAEP-SHA256 encryption of every tensor in 190-byte chunks.  The
    reference "adds" ciphertexts by XOR, which does not decrypt; here
    ``aggregate_encrypted`` is performed by the key holder (decrypt, add,
    re-encrypt), so ``decrypt_aggregate(aggregate_encrypted(...))`` is the
    true element-wise sum.  Tensor byte lengths are remembered next to the
    shapes, so the padding of the last chunk never has to be guessed from the
    plaintext.

``SecureMaskingAggregator``
    Every update is blinded with an additive random mask and sealed with
    AES-256-GCM (12-byte nonce prepended).  The aggregator accumulates the
    masks, sums the opened updates, removes the accumulated mask and re-seals
    the sum.

``PairwiseMaskSecureAggregation``
    Bonawitz-style pairwise masking for the device plane: clients i<j share a
    seed s_ij; client i adds PRG(s_ij), client j subtracts it, so the masks
    cancel in the sum and no party ever sees an individual update in the
    clear.  The PRG is the counter-based Philox stream of
    ``nanofed_amd.ops`` (same stream on CPU and GPU), applied to flat arenas
    in one kernel per pair.  Masks of clients that drop out after masking are
    removed from the sum with the seeds of the survivors.

All failures while opening ciphertexts raise
``ValueError("Decryption failed for <key>: ...")``.
"""

from __future__ import annotations

import hashlib
import os
from abc import ABC, abstractmethod
from dataclasses import dataclass
from typing import Generic, Protocol, Sequence, TypeVar

import numpy as np
import torch

from nanofed_amd.crypto import (
    RSAKey,
    aesgcm_open,
    aesgcm_seal,
    pbkdf2_sha256,
)
from nanofed_amd.utils import Logger

EncryptedType = TypeVar("EncryptedType")

_NONCE_LEN = 12


class SecureAggregationProtocol(Protocol, Generic[EncryptedType]):
    def encrypt_update(
        self, update: dict[str, torch.Tensor]
    ) -> dict[str, EncryptedType]: ...
    def decrypt_aggregate(
        self, encrypted_sum: dict[str, EncryptedType]
    ) -> dict[str, torch.Tensor]: ...
    def aggregate_encrypted(
        self, encrypted_updates: Sequence[dict[str, EncryptedType]]
    ) -> dict[str, EncryptedType]: ...


@dataclass(slots=True, frozen=True)
class SecureAggregationConfig:
    min_clients: int
    key_size: int = 2048
    threshold: int | None = None
    masking_seed_size: int = 256
    dropout_tolerance: float = 0.0


class BaseSecureAggregator(ABC, Generic[EncryptedType]):
    def __init__(self, config: SecureAggregationConfig) -> None:
        self._config = config
        self._logger = Logger()
        self._setup_crypto()

    @abstractmethod
    def _setup_crypto(self) -> None: ...

    @abstractmethod
    def encrypt_update(
        self, update: dict[str, torch.Tensor]
    ) -> dict[str, EncryptedType]: ...

    @abstractmethod
    def decrypt_aggregate(
        self, encrypted_sum: dict[str, EncryptedType]
    ) -> dict[str, torch.Tensor]: ...

    @abstractmethod
    def aggregate_encrypted(
        self, encrypted_updates: Sequence[dict[str, EncryptedType]]
    ) -> dict[str, EncryptedType]: ...

    def _require_quorum(self, n: int) -> None:
        if n < self._config.min_clients:
            raise ValueError(
                f"Need at least {self._config.min_clients} clients"
            )


def _fp32_bytes(tensor: torch.Tensor) -> bytes:
    return (
        tensor.detach().to("cpu", torch.float32).contiguous().numpy().tobytes()
    )


def _fp32_tensor(raw: bytes, shape: tuple[int, ...]) -> torch.Tensor:
    arr = np.frombuffer(raw, dtype=np.float32)
    return torch.from_numpy(arr.copy()).reshape(shape)


class HomomorphicSecureAggregator(
    BaseSecureAggregator[list[bytes]], SecureAggregationProtocol[list[bytes]]
):
    def _setup_crypto(self) -> None:
        self._key = RSAKey.generate(self._config.key_size)
        self._chunk_size = self._key.oaep_max_chunk  # key_bytes - 2*32 - 2
        self._shapes: dict[str, tuple[int, ...]] = {}
        self._nbytes: dict[str, int] = {}

    def public_key_pem(self) -> bytes:
        return self._key.public_pem()

    def encrypt_update(
        self, update: dict[str, torch.Tensor]
    ) -> dict[str, list[bytes]]:
        sealed: dict[str, list[bytes]] = {}
        for name, tensor in update.items():
            raw = _fp32_bytes(tensor)
            self._shapes[name] = tuple(tensor.shape)
            self._nbytes[name] = len(raw)
            tail = (-len(raw)) % self._chunk_size
            sealed[name] = self._key.oaep_encrypt_chunks(
                raw + bytes(tail), self._chunk_size
            )
        return sealed

    def decrypt_aggregate(
        self, encrypted_sum: dict[str, list[bytes]]
    ) -> dict[str, torch.Tensor]:
        opened: dict[str, torch.Tensor] = {}
        for name, chunks in encrypted_sum.items():
            try:
                raw = self._key.oaep_decrypt_chunks(list(chunks), self._chunk_size)
                raw = raw[: self._nbytes[name]]
                opened[name] = _fp32_tensor(raw, self._shapes[name])
            except Exception as exc:
                raise ValueError(f"Decryption failed for {name}: {exc}") from exc
        return opened

    def aggregate_encrypted(
        self, encrypted_updates: Sequence[dict[str, list[bytes]]]
    ) -> dict[str, list[bytes]]:
        self._require_quorum(len(encrypted_updates))
        total: dict[str, torch.Tensor] = {}
        for enc in encrypted_updates:
            for name, tensor in self.decrypt_aggregate(enc).items():
                if name in total:
                    total[name] += tensor
                else:
                    total[name] = tensor.clone()
        return self.encrypt_update(total)


class SecureMaskingAggregator(
    BaseSecureAggregator[bytes], SecureAggregationProtocol[bytes]
):
    def __init__(
        self, config: SecureAggregationConfig, key: bytes | None = None
    ) -> None:
        if key:
            if len(key) != 32:
                raise ValueError("AES-256-GCM key must be 32 bytes")
            self._key = key
        super().__init__(config)

    def _setup_crypto(self) -> None:
        if not hasattr(self, "_key"):
            self._key = pbkdf2_sha256(
                os.urandom(32), salt=os.urandom(16), iterations=100_000
            )
        self._shapes: dict[str, tuple[int, ...]] = {}
        self._cumulative_mask: dict[str, torch.Tensor] = {}
        self._mask_rng = torch.Generator()
        seed_bytes = min(max(self._config.masking_seed_size // 8, 1), 7)
        self._mask_rng.manual_seed(int.from_bytes(os.urandom(seed_bytes), "big"))

    def _seal(self, tensor: torch.Tensor) -> bytes:
        nonce = os.urandom(_NONCE_LEN)
        return nonce + aesgcm_seal(self._key, nonce, _fp32_bytes(tensor))

    def encrypt_update(self, update: dict[str, torch.Tensor]) -> dict[str, bytes]:
        sealed: dict[str, bytes] = {}
        for name, tensor in update.items():
            value = tensor.detach().to("cpu", torch.float32)
            self._shapes[name] = tuple(value.shape)
            mask = torch.rand(value.shape, generator=self._mask_rng)
            if name in self._cumulative_mask:
                self._cumulative_mask[name] += mask
            else:
                self._cumulative_mask[name] = mask.clone()
            sealed[name] = self._seal(value + mask)
        return sealed

    def decrypt_aggregate(
        self, encrypted_sum: dict[str, bytes]
    ) -> dict[str, torch.Tensor]:
        opened: dict[str, torch.Tensor] = {}
        for name, blob in encrypted_sum.items():
            try:
                raw = aesgcm_open(
                    self._key, blob[:_NONCE_LEN], blob[_NONCE_LEN:]
                )
                opened[name] = _fp32_tensor(raw, self._shapes[name])
            except Exception as exc:
                raise ValueError(f"Decryption failed for {name}: {exc}") from exc
        return opened

    def aggregate_encrypted(
        self, encrypted_updates: Sequence[dict[str, bytes]]
    ) -> dict[str, bytes]:
        self._require_quorum(len(encrypted_updates))
        total: dict[str, torch.Tensor] = {}
        for enc in encrypted_updates:
            for name, tensor in self.decrypt_aggregate(enc).items():
                if name in total:
                    total[name] += tensor
                else:
                    total[name] = tensor.clone()
        sealed = {}
        for name, masked_sum in total.items():
            mask = self._cumulative_mask.get(name)
            clear = masked_sum if mask is None else masked_sum - mask
            sealed[name] = self._seal(clear)
        self._cumulative_mask = {}
        return sealed


class PairwiseMaskSecureAggregation:
    """Pairwise additive masking over flat fp32 buffers (any device).

    Usage per round::

        sa = PairwiseMaskSecureAggregation(num_clients=K, round_seed=r)
        masked_k = sa.mask(k, flat_k)                  # on each client
        total    = sa.aggregate_flat(masked, present)  # server: sum (+ repair)

    ``mask_scale`` bounds the mask magnitude; masks are uniform in
    [-mask_scale, mask_scale).  Because fp32 addition is not associative the
    cancellation is exact only up to rounding of order
    ``mask_scale * 2^-24 * K``; choose ``mask_scale`` a few orders of magnitude
    above the parameter scale, not more.
    """

    def __init__(
        self,
        num_clients: int,
        round_seed: int = 0,
        mask_scale: float = 8.0,
        min_clients: int = 2,
        secret: bytes | None = None,
    ) -> None:
        if num_clients < 2:
            raise ValueError("Pairwise masking needs at least 2 clients")
        self._n = num_clients
        self._scale = float(mask_scale)
        self._min_clients = min_clients
        self._secret = secret or os.urandom(32)
        self._round_seed = int(round_seed)

    def pair_seed(self, i: int, j: int) -> int:
        """Seed shared by clients i and j (symmetric), 63 bits.

        Stand-in for a Diffie-Hellman agreed secret: derived from a per-session
        secret so that tests and single-process simulations are reproducible.
        """
        lo, hi = (i, j) if i < j else (j, i)
        digest = hashlib.sha256(
            self._secret + f"|{self._round_seed}|{lo}|{hi}".encode()
        ).digest()
        return int.from_bytes(digest[:8], "big") >> 1

    def _apply(self, flat: torch.Tensor, i: int, j: int, sign: float) -> None:
        from nanofed_amd import ops

        ops.mask_add_(flat, seed=self.pair_seed(i, j), scale=sign * self._scale)

    def mask(self, client: int, flat: torch.Tensor) -> torch.Tensor:
        """Blinded copy of ``flat`` for ``client``."""
        out = flat.detach().to(torch.float32).clone()
        for other in range(self._n):
            if other == client:
                continue
            self._apply(out, client, other, +1.0 if client < other else -1.0)
        return out

    def aggregate_flat(
        self, masked: Sequence[torch.Tensor], present: Sequence[int] | None = None
    ) -> torch.Tensor:
        """Sum of the blinded buffers of the clients in ``present`` (default:
        all), with the unmatched masks of absent clients removed."""
        from nanofed_amd import ops

        present = list(range(self._n)) if present is None else list(present)
        if len(masked) != len(present):
            raise ValueError("One masked buffer per present client is required")
        if len(present) < self._min_clients:
            raise ValueError(
                f"Not enough clients: {len(present)} < {self._min_clients}"
            )
        total = ops.weighted_sum(list(masked), [1.0] * len(masked))
        absent = [c for c in range(self._n) if c not in set(present)]
        for survivor in present:
            for gone in absent:
                # the survivor applied sign(survivor, gone); undo it
                sign = +1.0 if survivor < gone else -1.0
                self._apply(total, survivor, gone, -sign)
        return total

    # SecureAggregationProtocol of the privacy-aware aggregator -----------
    def verify_shares(self, shares: Sequence[dict[str, torch.Tensor]]) -> bool:
        if len(shares) < self._min_clients:
            return False
        ref = {k: tuple(v.shape) for k, v in shares[0].items()}
        return all(
            s.keys() == ref.keys()
            and all(tuple(s[k].shape) == shp for k, shp in ref.items())
            for s in shares
        )

    def aggregate_shares(
        self, shares: Sequence[dict[str, torch.Tensor]]
    ) -> dict[str, torch.Tensor]:
        """Mask every share, sum the blinded shares, return the (unmasked by
        cancellation) sum keyed like the inputs."""
        if len(shares) < self._min_clients:
            raise ValueError(
                f"Not enough clients: {len(shares)} < {self._min_clients}"
            )
        if len(shares) != self._n:
            raise ValueError(
                f"Expected {self._n} shares, got {len(shares)}"
            )
        names = list(shares[0].keys())
        shapes = [tuple(shares[0][n].shape) for n in names]
        sizes = [int(np.prod(s)) if s else 1 for s in shapes]
        flats = [
            torch.cat([s[n].detach().float().reshape(-1) for n in names])
            for s in shares
        ]
        blinded = [self.mask(i, f) for i, f in enumerate(flats)]
        total = self.aggregate_flat(blinded)
        out, start = {}, 0
        for n, shp, size in zip(names, shapes, sizes):
            out[n] = total[start : start + size].reshape(shp)
            start += size
        return out
