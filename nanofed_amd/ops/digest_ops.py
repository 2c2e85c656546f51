"""Tree digests of arena rows, for signing and verifying client updates on the device
plane.

The reference signs the serialised update on the host
(nanofed/server/validation.py:138-212: RSA-PSS over the tensors' bytes).  Moving
every client's row to the host for that would cost more than the round itself, so a row
is digested where it lies: the device hashes fixed-size chunks with SHA-256
(csrc/hip/digest_ops.hip, one thread per chunk), only the 32-byte chunk digests travel,
and the host hashes their concatenation (with a domain tag, the row length and the chunk
size) into the 32-byte root that gets signed.  The CPU path computes the same tree with
``hashlib``, so digests agree across devices.
"""

from __future__ import annotations

import hashlib
from ctypes import c_int, c_long, c_void_p

import torch

from nanofed_amd.core.exceptions import DeviceError

from . import library

_TAG = b"nanofed-amd/row-digest/v1"


def chunk_bytes_for(row_bytes: int) -> int:
    """1 KiB leaves for small models (many parallel leaves), 64 KiB beyond 16 MiB rows (so
    that the leaf digests stay a small fraction of the row)."""
    return 1024 if row_bytes <= (16 << 20) else 65536


def symbols() -> list[str]:
    return ["nf_sha256_chunks"]


def chunk_digests(rows: torch.Tensor, chunk_bytes: int | None = None) -> torch.Tensor:
    """uint8 ``[rows, chunks, 32]``: SHA-256 of every ``chunk_bytes`` slice of every row of a
    float32 ``[rows, n]`` tensor (unit column stride; rows may be strided)."""
    if rows.dim() != 2 or rows.dtype != torch.float32 or rows.stride(1) != 1:
        raise DeviceError("chunk_digests needs a float32 [rows, n] tensor with contiguous rows")
    G, n = rows.shape
    row_bytes = n * 4
    if row_bytes % 64:
        raise DeviceError("rows must be a multiple of 16 floats long")
    cb = chunk_bytes or chunk_bytes_for(row_bytes)
    chunks = -(-row_bytes // cb)
    if rows.device.type == "cuda":
        out = torch.empty(G, chunks, 32, dtype=torch.uint8, device=rows.device)
        fn = library.declare("nf_sha256_chunks",
                             [c_void_p, c_long, c_long, c_int, c_int, c_void_p, c_void_p], c_int)
        library.check(
            fn(c_void_p(rows.data_ptr()), row_bytes, rows.stride(0) * 4, G, cb,
               c_void_p(out.data_ptr()), library.stream_ptr(rows.device)),
            "sha256_chunks",
        )
        return out
    out = torch.empty(G, chunks, 32, dtype=torch.uint8)
    for g in range(G):
        raw = rows[g].detach().contiguous().numpy().tobytes()
        for c in range(chunks):
            d = hashlib.sha256(raw[c * cb : (c + 1) * cb]).digest()
            out[g, c] = torch.frombuffer(bytearray(d), dtype=torch.uint8)
    return out


def row_digests(rows: torch.Tensor, chunk_bytes: int | None = None) -> list[bytes]:
    """The 32-byte tree digest of every row (one device-to-host copy of the leaf digests)."""
    row_bytes = rows.shape[1] * 4
    cb = chunk_bytes or chunk_bytes_for(row_bytes)
    leaves = chunk_digests(rows, cb).cpu().numpy()
    head = _TAG + row_bytes.to_bytes(8, "big") + cb.to_bytes(8, "big")
    return [hashlib.sha256(head + leaves[g].tobytes()).digest() for g in range(rows.shape[0])]
