"""Causal self-attention on the packed ``[B*T, 3C]`` q|k|v layout
(csrc/hip/attention.hip: MFMA flash attention, head dimension 64).

Device tensors run the gfx950 kernels; CPU tensors run PyTorch's
``scaled_dot_product_attention`` in fp32 (and autograd for the backward),
which is also the oracle of the kernel tests.  The reference has no attention
layer; this serves the GPT-2 configuration of the benchmark suite.
"""

from __future__ import annotations

from ctypes import c_int, c_void_p

import torch
import torch.nn.functional as F

from nanofed_amd.core.exceptions import DeviceError

from . import library

P, I = c_void_p, c_int
_SIGS = {
    "nf_attn_seq_multiple": ([], I),
    "nf_attn_fwd": ([P, P, P, I, I, I, P], I),
    "nf_attn_bwd": ([P, P, P, P, P, P, I, I, I, P], I),
    "nf_attn_set_split": ([I], None),
}
_fns: dict[str, object] = {}
HEAD_DIM = 64


def _fn(name: str):
    fn = _fns.get(name)
    if fn is None:
        argtypes, restype = _SIGS[name]
        fn = library.declare(name, argtypes, restype)
        _fns[name] = fn
    return fn


def symbols() -> list[str]:
    return sorted(_SIGS)


def _p(t: torch.Tensor) -> c_void_p:
    return c_void_p(t.data_ptr())


def _check(qkv: torch.Tensor, batch: int, seq: int, heads: int) -> int:
    C = heads * HEAD_DIM
    if qkv.dtype != torch.bfloat16 or not qkv.is_contiguous():
        raise DeviceError("qkv must be a contiguous bf16 matrix")
    if tuple(qkv.shape) != (batch * seq, 3 * C):
        raise DeviceError(f"qkv must be [{batch * seq}, {3 * C}], got {tuple(qkv.shape)}")
    if qkv.device.type == "cuda":
        mult = int(_fn("nf_attn_seq_multiple")())
        if seq % mult:
            raise DeviceError(f"sequence length {seq} is not a multiple of {mult}")
    return C


def set_forward_split(groups: int) -> None:
    """Wave groups (1, 2 or 4) that share one 128-query tile in the forward
    kernel, each taking every ``groups``-th key stage (tuning hook)."""
    if groups not in (1, 2, 4):
        raise DeviceError("the forward split must be 1, 2 or 4")
    _fn("nf_attn_set_split")(groups)


def backward_scratch(batch: int, heads: int, seq: int, device: torch.device) -> torch.Tensor:
    return torch.empty(batch * heads * seq, dtype=torch.float32, device=device)


def _split(qkv: torch.Tensor, batch: int, seq: int, heads: int):
    q, k, v = qkv.float().view(batch, seq, 3, heads, HEAD_DIM).unbind(2)
    return q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)


def causal_attention_fwd(qkv: torch.Tensor, out: torch.Tensor, lse: torch.Tensor, batch: int,
                         seq: int, heads: int) -> None:
    """``out[B*T, C] = softmax(q k^T / 8 + causal mask) v`` per head;
    ``lse[B, H, T]`` receives the log-sum-exp of every score row."""
    C = _check(qkv, batch, seq, heads)
    if qkv.device.type == "cuda":
        library.check(_fn("nf_attn_fwd")(_p(qkv), _p(out), _p(lse), batch, seq, heads,
                                         library.stream_ptr(qkv.device)), "attn_fwd")
        return
    q, k, v = _split(qkv, batch, seq, heads)
    scores = (q @ k.transpose(-1, -2)) * HEAD_DIM**-0.5
    mask = torch.ones(seq, seq, dtype=torch.bool).tril()
    scores = scores.masked_fill(~mask, float("-inf"))
    lse.view(batch, heads, seq).copy_(torch.logsumexp(scores, -1))
    y = torch.softmax(scores, -1) @ v
    out.copy_(y.transpose(1, 2).reshape(batch * seq, C).to(torch.bfloat16))


def causal_attention_bwd(qkv: torch.Tensor, out: torch.Tensor, lse: torch.Tensor,
                         dout: torch.Tensor, dqkv: torch.Tensor, scratch: torch.Tensor,
                         batch: int, seq: int, heads: int) -> None:
    """``dqkv`` (same packed layout as ``qkv``, fully overwritten) from ``dout``."""
    C = _check(qkv, batch, seq, heads)
    if qkv.device.type == "cuda":
        library.check(_fn("nf_attn_bwd")(_p(qkv), _p(out), _p(lse), _p(dout), _p(dqkv),
                                         _p(scratch), batch, seq, heads,
                                         library.stream_ptr(qkv.device)), "attn_bwd")
        return
    with torch.enable_grad():
        leaf = qkv.float().detach().requires_grad_(True)
        q, k, v = leaf.view(batch, seq, 3, heads, HEAD_DIM).unbind(2)
        y = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2),
                                           v.transpose(1, 2), is_causal=True)
        y = y.transpose(1, 2).reshape(batch * seq, C)
        (g,) = torch.autograd.grad(y, leaf, dout.float())
    dqkv.copy_(g.to(torch.bfloat16))
