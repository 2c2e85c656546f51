"""Transformer building blocks (csrc/hip/nn_ops.hip) behind torch-tensor handles.

bf16 activations, fp32 statistics.  Device tensors go to the gfx950 kernels;
CPU tensors run the same arithmetic in PyTorch (rounding to bf16 at the same
points), which is also what the kernel tests compare against.  The reference
has no sequence model (its only network is nanofed/models/mnist.py:6-28); these
ops serve the GPT-2 configuration of the benchmark suite.
"""

from __future__ import annotations

import math
from ctypes import c_float, c_int, c_long, c_void_p

import torch
import torch.nn.functional as F

from nanofed_amd.core.exceptions import DeviceError

from . import library

P, L, I, FL = c_void_p, c_long, c_int, c_float

_SIGS = {
    "nf_ln_max_cols": ([], I),
    "nf_embed_fwd": ([P, P, P, P, I, I, I, P], I),
    "nf_embed_bwd": ([P, P, P, P, P, I, I, I, P], I),
    "nf_ln_fwd": ([P, P, P, P, P, P, P, P, I, I, FL, P], I),
    "nf_ln_bwd_blocks": ([I], I),
    "nf_ln_bwd": ([P, P, P, P, P, P, P, P, P, P, P, I, I, P], I),
    "nf_gelu_bwd_colsum": ([P, P, P, P, P, I, I, P], I),
    "nf_gelu_fwd": ([P, P, L, P], I),
    "nf_gelu_bwd": ([P, P, P, L, P], I),
    "nf_colsum_chunks": ([], I),
    "nf_colsum": ([P, P, P, I, I, L, P], I),
    "nf_ce_fwd_bwd": ([P, P, P, P, I, I, L, FL, P], I),
    "nf_ln_fwd_g": ([P, P, P, P, P, P, P, P, I, I, I, FL, L, P], I),
    "nf_ln_bwd_g": ([P, P, P, P, P, P, P, P, P, P, P, I, I, I, L, L, P], I),
    "nf_colsum_g": ([P, P, P, I, I, I, L, L, P], I),
    "nf_gelu_bwd_colsum_g": ([P, P, P, P, P, I, I, I, L, P], I),
    "nf_sgd_step_bf16g": ([P, P, P, L, FL, FL, P], I),
}

_fns: dict[str, object] = {}


def _fn(name: str):
    fn = _fns.get(name)
    if fn is None:
        argtypes, restype = _SIGS[name]
        fn = library.declare(name, argtypes, restype)
        _fns[name] = fn
    return fn


def symbols() -> list[str]:
    return sorted(_SIGS)


def _p(t: torch.Tensor | None) -> c_void_p:
    return c_void_p(t.data_ptr() if t is not None else 0)


def _need(t: torch.Tensor, dtype: torch.dtype, name: str) -> None:
    if t.dtype != dtype:
        raise DeviceError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise DeviceError(f"{name} must be contiguous")


def _gpu(t: torch.Tensor) -> bool:
    return t.device.type == "cuda"


_ws: dict[tuple, torch.Tensor] = {}


def _workspace(device: torch.device, floats: int) -> torch.Tensor:
    key = (device.index or 0, torch.cuda.current_stream(device).cuda_stream)
    ws = _ws.get(key)
    if ws is None or ws.numel() < floats:
        ws = torch.empty(max(floats, 1 << 20), dtype=torch.float32, device=device)
        _ws[key] = ws
    return ws


# ---------------------------------------------------------------------------
def embed_fwd(tokens: torch.Tensor, wte: torch.Tensor, wpe: torch.Tensor, seq_len: int,
              out: torch.Tensor | None = None) -> torch.Tensor:
    """``x[r] = wte[tokens[r]] + wpe[r % seq_len]`` for flat ``tokens`` ``[R]``."""
    _need(tokens, torch.int64, "tokens")
    _need(wte, torch.bfloat16, "wte")
    _need(wpe, torch.bfloat16, "wpe")
    R, C = tokens.numel(), wte.shape[1]
    if out is None:
        out = torch.empty(R, C, dtype=torch.bfloat16, device=wte.device)
    if _gpu(wte):
        library.check(_fn("nf_embed_fwd")(_p(tokens), _p(wte), _p(wpe), _p(out), R, seq_len, C,
                                          library.stream_ptr(wte.device)), "embed_fwd")
    else:
        pos = torch.arange(R, device=tokens.device) % seq_len
        out.copy_((wte[tokens.reshape(-1)].float() + wpe[pos].float()).to(torch.bfloat16))
    return out


def embed_bwd(tokens: torch.Tensor, dx: torch.Tensor, dwte: torch.Tensor, dwpe: torch.Tensor,
              batch: int, seq_len: int, sorted_tokens: torch.Tensor | None = None,
              order: torch.Tensor | None = None) -> None:
    """``dwte += scatter(dx by token)`` (in token-sorted order, no atomics) and
    ``dwpe = sum over the batch``.  ``sorted_tokens`` / ``order``: the stable sort of the
    flat tokens when the caller has it already (``token_order``: drawn on the host with
    the batch, so that a captured step holds no sort)."""
    _need(dx, torch.bfloat16, "dx")
    _need(dwte, torch.bfloat16, "dwte")
    _need(dwpe, torch.bfloat16, "dwpe")
    C = dx.shape[-1]
    flat = tokens.reshape(-1)
    if _gpu(dx):
        if sorted_tokens is None or order is None:
            srt, order = torch.sort(flat, stable=True)
        else:
            srt = sorted_tokens
            for t, name in ((srt, "sorted_tokens"), (order, "order")):
                _need(t, torch.int64, name)
                if t.numel() != flat.numel():
                    raise DeviceError(f"embed_bwd: {name} must have one entry per token")
        library.check(_fn("nf_embed_bwd")(_p(srt), _p(order), _p(dx), _p(dwte), _p(dwpe), batch,
                                          seq_len, C, library.stream_ptr(dx.device)), "embed_bwd")
    else:
        acc = dwte.float()
        acc.index_add_(0, flat, dx.reshape(-1, C).float())
        dwte.copy_(acc.to(torch.bfloat16))
        dwpe[:seq_len].copy_(dx.reshape(batch, seq_len, C).float().sum(0).to(torch.bfloat16))


def token_order(tokens_host: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """``(sorted tokens, order)`` of a host token tensor, flattened: what
    ``torch.sort(flat, stable=True)`` returns, computed with numpy on the host."""
    import numpy as np

    flat = tokens_host.reshape(-1).numpy()
    order = np.argsort(flat, kind="stable")
    return torch.from_numpy(flat[order]), torch.from_numpy(order.astype(np.int64))


def ln_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, y: torch.Tensor,
           mean: torch.Tensor, rstd: torch.Tensor, delta: torch.Tensor | None = None,
           x_out: torch.Tensor | None = None, eps: float = 1e-5) -> None:
    """``y = LayerNorm(x)``; with ``delta``: ``x_out = x + delta`` (rounded to
    bf16) and ``y = LayerNorm(x_out)``."""
    for t, n in ((x, "x"), (gamma, "gamma"), (beta, "beta"), (y, "y")):
        _need(t, torch.bfloat16, n)
    R, C = x.shape
    if delta is not None and x_out is None:
        raise DeviceError("ln_fwd with delta needs x_out")
    if _gpu(x):
        library.check(_fn("nf_ln_fwd")(_p(x), _p(delta), _p(x_out), _p(gamma), _p(beta), _p(y),
                                       _p(mean), _p(rstd), R, C, eps,
                                       library.stream_ptr(x.device)), "ln_fwd")
        return
    xv = x.float()
    if delta is not None:
        xv = (xv + delta.float()).to(torch.bfloat16)
        x_out.copy_(xv)
        xv = xv.float()
    mu = xv.mean(1)
    var = ((xv - mu[:, None]) ** 2).mean(1)
    rs = torch.rsqrt(var + eps)
    mean.copy_(mu)
    rstd.copy_(rs)
    y.copy_(((xv - mu[:, None]) * rs[:, None] * gamma.float() + beta.float()).to(torch.bfloat16))


def ln_bwd(dy: torch.Tensor, x: torch.Tensor, gamma: torch.Tensor, mean: torch.Tensor,
           rstd: torch.Tensor, dx: torch.Tensor, dgamma: torch.Tensor, dbeta: torch.Tensor,
           dres: torch.Tensor | None = None, dxsum: torch.Tensor | None = None) -> None:
    """``dx = LayerNorm'(dy) (+ dres)``, ``dgamma``, ``dbeta`` (bf16, overwritten);
    ``dxsum`` (optional) receives the column sums of ``dx``."""
    R, C = x.shape
    if _gpu(x):
        nb = int(_fn("nf_ln_bwd_blocks")(R))
        ws = _workspace(x.device, nb * 3 * C)
        library.check(_fn("nf_ln_bwd")(_p(dy), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dres),
                                       _p(dx), _p(dgamma), _p(dbeta), _p(dxsum), _p(ws), R, C,
                                       library.stream_ptr(x.device)), "ln_bwd")
        return
    xh = (x.float() - mean[:, None]) * rstd[:, None]
    d = dy.float()
    dyw = d * gamma.float()
    m1 = dyw.mean(1, keepdim=True)
    m2 = (dyw * xh).mean(1, keepdim=True)
    out = (dyw - m1 - xh * m2) * rstd[:, None]
    if dres is not None:
        out = out + dres.float()
    dgamma.copy_((d * xh).sum(0).to(torch.bfloat16))
    dbeta.copy_(d.sum(0).to(torch.bfloat16))
    rounded = out.to(torch.bfloat16)
    if dxsum is not None:
        # column sums of the gradient as it is handed on (bf16), like the kernel
        dxsum.copy_(rounded.float().sum(0).to(torch.bfloat16))
    dx.copy_(rounded)


def gelu_fwd(x: torch.Tensor, y: torch.Tensor) -> None:
    if _gpu(x):
        _need(x, torch.bfloat16, "x")
        library.check(_fn("nf_gelu_fwd")(_p(x), _p(y), x.numel(), library.stream_ptr(x.device)),
                      "gelu_fwd")
    else:
        y.copy_(F.gelu(x.float(), approximate="tanh").to(torch.bfloat16))


def gelu_bwd(x: torch.Tensor, dy: torch.Tensor, dx: torch.Tensor) -> None:
    if _gpu(x):
        library.check(_fn("nf_gelu_bwd")(_p(x), _p(dy), _p(dx), x.numel(),
                                         library.stream_ptr(x.device)), "gelu_bwd")
    else:
        v = x.float()
        k, c = math.sqrt(2.0 / math.pi), 0.044715
        t = torch.tanh(k * (v + c * v**3))
        dt = (1 - t * t) * k * (1 + 3 * c * v * v)
        dx.copy_((dy.float() * (0.5 * (1 + t) + 0.5 * v * dt)).to(torch.bfloat16))


def gelu_bwd_colsum(x: torch.Tensor, dy: torch.Tensor, dx: torch.Tensor,
                    dbias: torch.Tensor) -> None:
    """``dx = dy * gelu'(x)`` and ``dbias = column sums of dx`` in one pass over
    contiguous ``[R, N]`` matrices."""
    R, N = x.shape
    if _gpu(x):
        _need(x, torch.bfloat16, "x")
        _need(dy, torch.bfloat16, "dy")
        chunks = int(_fn("nf_colsum_chunks")())
        ws = _workspace(x.device, chunks * N)
        library.check(_fn("nf_gelu_bwd_colsum")(_p(x), _p(dy), _p(dx), _p(dbias), _p(ws), R, N,
                                                library.stream_ptr(x.device)), "gelu_bwd_colsum")
    else:
        gelu_bwd(x, dy, dx)
        dbias.copy_(dx.float().sum(0).to(torch.bfloat16))


def colsum(m: torch.Tensor, out: torch.Tensor) -> None:
    """``out[c] = sum_r m[r, c]`` for a bf16 matrix whose rows may be strided."""
    R, N = m.shape
    if m.stride(1) != 1:
        raise DeviceError("colsum needs unit column stride")
    if _gpu(m):
        chunks = int(_fn("nf_colsum_chunks")())
        ws = _workspace(m.device, chunks * N)
        library.check(_fn("nf_colsum")(_p(m), _p(out), _p(ws), R, N, m.stride(0),
                                       library.stream_ptr(m.device)), "colsum")
    else:
        out.copy_(m.float().sum(0).to(torch.bfloat16))


def cross_entropy_fwd_bwd(logits: torch.Tensor, targets: torch.Tensor, vocab: int,
                          scale: float, loss_rows: torch.Tensor,
                          correct_rows: torch.Tensor) -> None:
    """Per-row loss / top-1 hit of ``logits[:, :vocab]`` and, in place,
    ``logits = (softmax - onehot) * scale`` (padding columns become 0)."""
    _need(logits, torch.bfloat16, "logits")
    _need(targets, torch.int64, "targets")
    R, ld = logits.shape
    if _gpu(logits):
        library.check(_fn("nf_ce_fwd_bwd")(_p(logits), _p(targets), _p(loss_rows),
                                           _p(correct_rows), R, vocab, ld, scale,
                                           library.stream_ptr(logits.device)), "ce_fwd_bwd")
        return
    lg = logits[:, :vocab].float()
    lse = torch.logsumexp(lg, 1)
    loss_rows.copy_(lse - lg.gather(1, targets[:, None]).squeeze(1))
    correct_rows.copy_((lg.argmax(1) == targets).float())
    p = torch.exp(lg - lse[:, None])
    p[torch.arange(R), targets] -= 1.0
    logits.zero_()
    logits[:, :vocab].copy_((p * scale).to(torch.bfloat16))


def sgd_step_bf16g(p: torch.Tensor, g: torch.Tensor, shadow: torch.Tensor, lr: float,
                   weight_decay: float = 0.0) -> None:
    """``p -= lr * (g + wd * p)`` with bf16 gradients; ``shadow = bf16(p)``."""
    _need(p, torch.float32, "p")
    _need(g, torch.bfloat16, "g")
    _need(shadow, torch.bfloat16, "shadow")
    if _gpu(p):
        library.check(_fn("nf_sgd_step_bf16g")(_p(p), _p(g), _p(shadow), p.numel(), lr,
                                               weight_decay, library.stream_ptr(p.device)),
                      "sgd_step_bf16g")
    else:
        p.sub_(lr * (g.float() + weight_decay * p))
        shadow.copy_(p.to(torch.bfloat16))


# -- grouped variants: G clients' rows stacked in the matrices, per-client vectors at a fixed
#    distance from each other (rows of the bf16 parameter copy / the gradient buffer) ---------

def ln_fwd_g(x: torch.Tensor, gamma, beta, y: torch.Tensor, mean: torch.Tensor,
             rstd: torch.Tensor, G: int, delta: torch.Tensor | None = None,
             x_out: torch.Tensor | None = None, eps: float = 1e-5) -> None:
    """``ln_fwd`` for G groups of ``x.shape[0] / G`` rows; ``gamma`` / ``beta`` are
    ``group_ops.Ref``s: group g's vector lies ``stride`` elements after group g - 1's."""
    RT, C = x.shape
    if RT % G or gamma.stride != beta.stride:
        raise DeviceError("ln_fwd_g: rows must split evenly over the groups")
    library.check(_fn("nf_ln_fwd_g")(_p(x), _p(delta), _p(x_out), gamma.ptr, beta.ptr, _p(y),
                                     _p(mean), _p(rstd), G, RT // G, C, eps, gamma.stride,
                                     library.stream_ptr(x.device)), "ln_fwd_g")


def ln_bwd_g(dy: torch.Tensor, x: torch.Tensor, gamma, mean: torch.Tensor, rstd: torch.Tensor,
             dx: torch.Tensor, dgamma, dbeta, G: int, dres: torch.Tensor | None = None,
             dxsum=None) -> None:
    """``ln_bwd`` for G groups; ``gamma`` is a ``Ref`` into the parameter copy, ``dgamma``
    / ``dbeta`` / ``dxsum`` are ``Ref``s into one gradient buffer."""
    RT, C = x.shape
    if RT % G or dgamma.stride != dbeta.stride or (dxsum is not None and dxsum.stride != dgamma.stride):
        raise DeviceError("ln_bwd_g: inconsistent group layout")
    nb = int(_fn("nf_ln_bwd_blocks")(RT // G))
    ws = _workspace(x.device, G * nb * 3 * C)
    library.check(_fn("nf_ln_bwd_g")(_p(dy), _p(x), gamma.ptr, _p(mean), _p(rstd), _p(dres), _p(dx),
                                     dgamma.ptr, dbeta.ptr,
                                     dxsum.ptr if dxsum is not None else c_void_p(0), _p(ws), G,
                                     RT // G, C, gamma.stride, dgamma.stride,
                                     library.stream_ptr(x.device)), "ln_bwd_g")


def colsum_g(m: torch.Tensor, out, G: int) -> None:
    """Per-group column sums of a contiguous-row bf16 matrix into ``out`` (a ``Ref``)."""
    RT, N = m.shape
    if RT % G or m.stride(1) != 1:
        raise DeviceError("colsum_g: rows must split evenly over the groups")
    chunks = int(_fn("nf_colsum_chunks")())
    ws = _workspace(m.device, G * chunks * N)
    library.check(_fn("nf_colsum_g")(_p(m), out.ptr, _p(ws), G, RT // G, N, m.stride(0), out.stride,
                                     library.stream_ptr(m.device)), "colsum_g")


def gelu_bwd_colsum_g(x: torch.Tensor, dy: torch.Tensor, dx: torch.Tensor, dbias, G: int) -> None:
    """``gelu_bwd_colsum`` with one bias gradient per group (``dbias`` a ``Ref``)."""
    RT, N = x.shape
    if RT % G:
        raise DeviceError("gelu_bwd_colsum_g: rows must split evenly over the groups")
    _need(x, torch.bfloat16, "x")
    _need(dy, torch.bfloat16, "dy")
    chunks = int(_fn("nf_colsum_chunks")())
    ws = _workspace(x.device, G * chunks * N)
    library.check(_fn("nf_gelu_bwd_colsum_g")(_p(x), _p(dy), _p(dx), dbias.ptr, _p(ws), G, RT // G,
                                              N, dbias.stride, library.stream_ptr(x.device)),
                  "gelu_bwd_colsum_g")
