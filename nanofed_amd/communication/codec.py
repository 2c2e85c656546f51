"""Model-state codec for the HTTP wire format.

The wire schema (reference: nanofed/communication/http/types.py:14-50) carries
``model_state`` as ``{name: nested JSON lists}``.  The reference produces and
consumes it through Python float objects; for the 1.2M-parameter CNN that is
~25 MB of text and 1-2 s per transfer.  Here the tensors are converted
directly between fp32 buffers and JSON text by the native codec
(csrc/host/tensor_json.cpp), and the small envelope (ids, metrics, timestamps)
is still handled by the standard ``json`` module.  The text produced is plain
JSON that any reference client or server can parse.

``encode_state_json`` / ``decode_state_json`` work on the ``model_state``
object alone; ``dumps_envelope`` / ``loads_envelope`` splice it in and out of
a full request or response body.
"""

from __future__ import annotations

import ctypes
import json
from ctypes import POINTER, c_char_p, c_int, c_long, c_void_p
from functools import lru_cache
from typing import Any

import numpy as np
import torch

_STATE_KEY = '"model_state"'
_PLACEHOLDER = "__nanofed_model_state__"


@lru_cache(maxsize=1)
def _lib() -> ctypes.CDLL:
    from nanofed_amd import native_build

    if not native_build.HOST_LIB.exists():
        native_build.build_host()
    lib = ctypes.CDLL(str(native_build.HOST_LIB))
    lib.nf_json_encode_bound.restype = c_long
    lib.nf_json_encode_bound.argtypes = [c_long, c_int]
    lib.nf_json_encode_f32.restype = c_long
    lib.nf_json_encode_f32.argtypes = [
        c_void_p, POINTER(c_long), c_int, c_void_p, c_long]  # data, shape, ndim, out, cap
    lib.nf_json_decode_f32.restype = c_long
    lib.nf_json_decode_f32.argtypes = [
        c_void_p, c_long, c_void_p, c_long, POINTER(c_long), POINTER(c_int),
        POINTER(c_long)]
    lib.nf_json_skip_value.restype = c_long
    lib.nf_json_skip_value.argtypes = [c_char_p, c_long, c_long]
    return lib


def native_available() -> bool:
    try:
        _lib()
        return True
    except Exception:
        return False


def _as_f32_array(tensor: torch.Tensor) -> np.ndarray:
    arr = tensor.detach().to("cpu", torch.float32).numpy()
    # (ascontiguousarray alone would turn a 0-d tensor into a list of one number)
    return np.ascontiguousarray(arr).reshape(arr.shape)


def _encode_into(arr: np.ndarray, buf: np.ndarray, pos: int) -> int:
    """Write the JSON text of ``arr`` at ``buf[pos:]`` (uint8, room for the bound);
    returns the position after it."""
    lib = _lib()
    shape = (c_long * max(arr.ndim, 1))(*arr.shape)
    n = lib.nf_json_encode_f32(
        arr.ctypes.data_as(c_void_p), shape, arr.ndim,
        c_void_p(buf.ctypes.data + pos), buf.size - pos,
    )
    if n < 0:
        raise ValueError("JSON buffer too small")  # the caller sizes it from the bound
    return pos + int(n)


def _text_bound(arr: np.ndarray) -> int:
    """Room the text of ``arr`` certainly fits in: 17 bytes per number, 4 per list."""
    lists, numel = 0, 1
    for d in arr.shape:
        lists += numel
        numel *= d
    return numel * 17 + lists * 4 + 16


def encode_tensor_json(tensor: torch.Tensor) -> bytes:
    """Nested-list JSON text of ``tensor`` (as fp32; integers keep a ``.0``)."""
    arr = _as_f32_array(tensor)
    buf = np.empty(_text_bound(arr), dtype=np.uint8)  # not zero-filled
    end = _encode_into(arr, buf, 0)
    return buf[:end].tobytes()


def _decode_prefix(text: bytes, pos: int) -> tuple[torch.Tensor, int]:
    """Decode the JSON value starting at ``text[pos]``; returns (tensor, end).  The text
    is parsed in place (no scan for the value's end first, no copy of the value)."""
    lib = _lib()
    remaining = len(text) - pos
    if remaining <= 0:
        raise ValueError("Truncated JSON value in model_state")
    # a number takes at least 2 bytes ("0,") so len/2 bounds the value count (the buffer
    # is only touched where values land)
    cap = remaining // 2 + 2
    out = np.empty(cap, dtype=np.float32)
    shape = (c_long * 8)()
    ndim = c_int(0)
    consumed = c_long(0)
    base = ctypes.cast(ctypes.c_char_p(text), c_void_p).value
    n = lib.nf_json_decode_f32(
        c_void_p(base + pos), remaining, out.ctypes.data_as(c_void_p), cap, shape,
        ctypes.byref(ndim), ctypes.byref(consumed),
    )
    if n < 0 or n > cap:
        raise ValueError("Malformed tensor in model_state")
    dims = tuple(int(shape[i]) for i in range(ndim.value))
    tensor = torch.from_numpy(out[:n].copy()).reshape(dims)
    return tensor, pos + int(consumed.value)


def encode_state_json(state: dict[str, Any]) -> bytes:
    """``{"name": [[...]], ...}`` for a dict of tensors (or nested lists): written into one
    buffer, copied out once."""
    entries = []
    room = 2
    for name, value in state.items():
        key = json.dumps(str(name)).encode() + b": "
        body = _as_f32_array(value) if isinstance(value, torch.Tensor) else json.dumps(value).encode()
        entries.append((key, body))
        room += len(key) + 2 + (_text_bound(body) if isinstance(body, np.ndarray) else len(body))
    buf = np.empty(room, dtype=np.uint8)
    pos = 0

    def put(raw: bytes) -> None:
        nonlocal pos
        buf[pos : pos + len(raw)] = np.frombuffer(raw, dtype=np.uint8)
        pos += len(raw)

    put(b"{")
    for i, (key, body) in enumerate(entries):
        if i:
            put(b", ")
        put(key)
        if isinstance(body, np.ndarray):
            pos = _encode_into(body, buf, pos)
        else:
            put(body)
    put(b"}")
    return buf[:pos].tobytes()


def decode_state_json(text: bytes, start: int = 0, stop: int | None = None) -> dict[str, torch.Tensor]:
    """Inverse of :func:`encode_state_json` for list-valued states: the JSON object in
    ``text[start:stop]``, parsed in place."""
    n = len(text) if stop is None else stop
    pos = text.find(b"{", start, n)
    if pos < 0:
        raise ValueError("model_state is not a JSON object")
    pos += 1
    lib = _lib()
    state: dict[str, torch.Tensor] = {}
    while True:
        while pos < n and text[pos : pos + 1] in b" \n\t\r,":
            pos += 1
        if pos >= n:
            raise ValueError("Unterminated model_state object")
        if text[pos : pos + 1] == b"}":
            return state
        key_end = int(lib.nf_json_skip_value(text, n, pos))
        if key_end < 0 or text[pos : pos + 1] != b'"':
            raise ValueError("Malformed key in model_state")
        key = json.loads(text[pos:key_end])
        pos = text.find(b":", key_end, n) + 1
        if pos <= 0:
            raise ValueError("Missing ':' in model_state")
        tensor, pos = _decode_prefix(text, pos)
        if pos > n:
            raise ValueError("Truncated JSON value in model_state")
        state[key] = tensor


class EncodedState:
    """The JSON text of a ``model_state`` that has been encoded already (a server hands the
    same global model to every client of a round: one encoding, many responses)."""

    __slots__ = ("text",)

    def __init__(self, text: bytes) -> None:
        self.text = text


def state_fingerprint(state: dict[str, Any]) -> tuple:
    """Changes whenever a tensor of ``state`` is replaced or written in place (storage
    address, shape and PyTorch's in-place version counter of every entry)."""
    return tuple(
        (k, v.data_ptr(), tuple(v.shape), v._version) if isinstance(v, torch.Tensor) else (k, id(v))
        for k, v in state.items()
    )


def dumps_envelope(payload: dict[str, Any]) -> bytes:
    """JSON body for ``payload``; a tensor-valued ``model_state`` entry is
    encoded natively (or taken as it is from an ``EncodedState``), everything else with
    ``json``."""
    state = payload.get("model_state")
    if isinstance(state, EncodedState):
        encoded = state.text
    elif isinstance(state, dict) and native_available():
        encoded = encode_state_json(state)
    else:
        return json.dumps(_plain(payload)).encode()
    shell = dict(payload)
    shell["model_state"] = _PLACEHOLDER
    text = json.dumps(_plain(shell)).encode()
    return text.replace(json.dumps(_PLACEHOLDER).encode(), encoded, 1)


def loads_envelope(body: bytes) -> dict[str, Any]:
    """Parse a request/response body.  ``model_state`` comes back as a dict of
    fp32 tensors (or ``None``); the other fields as plain JSON values."""
    if not native_available():
        data = json.loads(body)
        if isinstance(data.get("model_state"), dict):
            data["model_state"] = {
                k: torch.tensor(v, dtype=torch.float32)
                for k, v in data["model_state"].items()
            }
        return data
    at = body.find(_STATE_KEY.encode())
    if at < 0:
        return json.loads(body)
    colon = body.find(b":", at + len(_STATE_KEY))
    lib = _lib()
    end = int(lib.nf_json_skip_value(body, len(body), colon + 1))
    if end < 0:
        raise ValueError("Truncated model_state")
    envelope = json.loads(body[: colon + 1] + b" null" + body[end:])
    first = colon + 1
    while first < end and body[first : first + 1] in b" \n\t\r":
        first += 1
    if body[first : first + 1] == b"{":
        envelope["model_state"] = decode_state_json(body, first, end)
    else:
        envelope["model_state"] = json.loads(body[colon + 1 : end])
    return envelope


def _plain(value: Any) -> Any:
    if isinstance(value, torch.Tensor):
        return value.detach().cpu().tolist()
    if isinstance(value, np.ndarray):
        return value.tolist()
    if isinstance(value, dict):
        return {k: _plain(v) for k, v in value.items()}
    if isinstance(value, (list, tuple)):
        return [_plain(v) for v in value]
    if hasattr(value, "__dataclass_fields__"):
        return {f: _plain(getattr(value, f)) for f in value.__dataclass_fields__}
    return value
