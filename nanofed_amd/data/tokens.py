"""Token streams from files on disk for the GPT-2 configuration (the reference ships a
loader only for the dataset of its own model, nanofed/data/mnist.py:9-40 - this is the same
entry for the language-model configuration).

Format: a flat array of token ids, the layout pre-tokenised corpora are commonly stored in -
``train.bin`` / ``val.bin`` as little-endian ``uint16`` (GPT-2's 50,257 ids fit) or ``uint32``,
or ``train.npy`` / ``val.npy`` holding a one-dimensional integer array.  Files are
memory-mapped, never loaded whole and never unpickled (``numpy.load`` with
``allow_pickle=False``); only the windows that are asked for are read.

The stream is cut into windows of ``seq_len + 1`` tokens that start every ``seq_len`` tokens
(inputs are a window's first ``seq_len`` tokens, targets its last ``seq_len``: every token of
the stream is predicted once); a seeded permutation picks the windows, so clients get
different, reproducible parts of the corpus.
"""

from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from nanofed_amd.utils import Logger

from .synthetic import synthetic_tokens

_DTYPES = {"uint16": np.dtype("<u2"), "uint32": np.dtype("<u4"), "int32": np.dtype("<i4"),
           "int64": np.dtype("<i8")}
_SPLIT_NAMES = {"train": ("train", "tokens_train"), "val": ("val", "validation", "test", "tokens_val")}


def find_token_file(data_dir: str | Path, split: str = "train") -> Path | None:
    """``<data_dir>/<split>.bin`` or ``.npy`` (``val`` also answers to ``validation`` /
    ``test``), or ``None``."""
    if split not in _SPLIT_NAMES:
        raise ValueError(f"split must be one of {sorted(_SPLIT_NAMES)}, not {split!r}")
    root = Path(data_dir)
    for stem in _SPLIT_NAMES[split]:
        for suffix in (".bin", ".npy"):
            path = root / (stem + suffix)
            if path.is_file():
                return path
    return None


def open_token_stream(path: str | Path, dtype: str = "uint16") -> np.ndarray:
    """The file as a read-only, memory-mapped one-dimensional array of token ids."""
    path = Path(path)
    if path.suffix == ".npy":
        arr = np.load(path, mmap_mode="r", allow_pickle=False)
        if arr.ndim != 1 or not np.issubdtype(arr.dtype, np.integer):
            raise ValueError(f"{path}: a one-dimensional integer array is required, "
                             f"found {arr.dtype} with shape {arr.shape}")
        return arr
    if dtype not in _DTYPES:
        raise ValueError(f"token dtype must be one of {sorted(_DTYPES)}, not {dtype!r}")
    dt = _DTYPES[dtype]
    size = path.stat().st_size
    if size == 0 or size % dt.itemsize:
        raise ValueError(f"{path}: {size} bytes is not a whole number of {dtype} tokens")
    return np.memmap(path, dtype=dt, mode="r")


def token_windows(stream: np.ndarray, seq_len: int, count: int, vocab_size: int, seed: int = 0,
                  source: str = "stream") -> torch.Tensor:
    """``int64 [count, seq_len + 1]``: ``count`` windows of the stream, chosen by a seeded
    permutation of the window starts ``0, seq_len, 2 seq_len, ...``.  Ids outside
    ``[0, vocab_size)`` are an error (they would index past the embedding matrix)."""
    if seq_len < 1 or count < 1:
        raise ValueError("seq_len and count must be positive")
    available = (len(stream) - 1) // seq_len
    if available < count:
        raise ValueError(
            f"{source}: {len(stream)} tokens hold {max(available, 0)} windows of {seq_len} + 1 "
            f"tokens, {count} are asked for")
    starts = np.sort(np.random.default_rng(seed).permutation(available)[:count]) * seq_len
    # (sorted: the rows are read in file order; which client gets which row is the
    # partition's business)
    out = np.empty((count, seq_len + 1), dtype=np.int64)
    for i, s in enumerate(starts):
        out[i] = stream[s : s + seq_len + 1]
    lo, hi = int(out.min()), int(out.max())
    if lo < 0 or hi >= vocab_size:
        raise ValueError(
            f"{source}: token ids span [{lo}, {hi}], the model's vocabulary has {vocab_size} "
            "entries (wrong token dtype, or another tokenizer?)")
    return torch.from_numpy(out)


def load_token_sequences(
    data_dir: str | Path,
    split: str,
    seq_len: int,
    count: int,
    vocab_size: int = 50257,
    seed: int = 0,
    dtype: str = "uint16",
    synthetic_fallback: bool = False,
) -> torch.Tensor:
    """``count`` training windows (``int64 [count, seq_len + 1]``) from the token file of
    ``split`` under ``data_dir``; with ``synthetic_fallback`` a missing file is replaced by
    the synthetic Markov stream and a warning."""
    path = find_token_file(data_dir, split)
    if path is None:
        if synthetic_fallback:
            Logger().warning(f"no {split}.bin / {split}.npy under {data_dir}; using synthetic tokens")
            return synthetic_tokens(count, seq_len, vocab_size, seed=seed, table_seed=0)
        raise RuntimeError(
            f"no token file for split {split!r} under {data_dir} (looked for "
            f"{', '.join(s + '.bin' for s in _SPLIT_NAMES[split])} and the same with .npy). "
            "Place one there or pass synthetic_fallback=True.")
    return token_windows(open_token_stream(path, dtype), seq_len, count, vocab_size, seed, str(path))


def write_token_file(path: str | Path, tokens: torch.Tensor | np.ndarray, dtype: str = "uint16") -> None:
    """Write a flat token array in the format ``open_token_stream`` reads (exports of other
    corpora, and the tests' fixtures)."""
    arr = np.asarray(tokens).reshape(-1)
    path = Path(path)
    if path.suffix == ".npy":
        np.save(path, arr.astype(_DTYPES.get(dtype, np.dtype("<i8"))), allow_pickle=False)
        return
    dt = _DTYPES[dtype]
    info = np.iinfo(dt)
    if arr.size and (int(arr.min()) < info.min or int(arr.max()) > info.max):
        raise ValueError(f"token ids do not fit {dtype}")
    arr.astype(dt).tofile(str(path))
