"""bf16 GEMMs on MFMA (csrc/hip/gemm_ops.hip) for the linear layers of the
transformer trainer (the reference's only linear layers are ``nn.Linear`` in
nanofed/models/mnist.py:24,27; its training step is nanofed/trainer/base.py:141-159).

Three operand layouts cover a linear layer's forward, data gradient and weight
gradient without any transposed copy:

``linear_fwd``    y[M, N]  = x[M, K] @ w[N, K]^T (+ bias) (+ GELU)      "NT"
``linear_dgrad``  dx[M, K] = dy[M, N] @ w[N, K]                         "NN"
``linear_wgrad``  dw[N, K] = dy[M, N]^T @ x[M, K]                        "TN"

Operands and results are bf16, accumulation is fp32.  CPU tensors (and GPUs
without the library, which cannot happen on a built tree) run the same
arithmetic through ``torch.mm`` in fp32 - that is also the oracle of the kernel
tests.
"""

from __future__ import annotations

import os

import torch

from . import library


def enabled(device: torch.device | None = None) -> bool:
    """True when GEMMs of ``device`` tensors run on the hand-written kernels
    (``NANOFED_NATIVE_GEMM=0`` routes them to the vendor library instead, for A/B
    measurements)."""
    if device is not None and torch.device(device).type != "cuda":
        return False
    if os.environ.get("NANOFED_NATIVE_GEMM", "1") == "0":
        return False
    try:
        return hasattr(library.hip_lib(), "nf_gemm_bf16")
    except Exception:  # noqa: BLE001 - no library: nothing native is enabled
        return False


_slabs: dict[tuple[int, int], torch.Tensor] = {}

NT, NN, TN = 0, 1, 2  # operand layouts (see csrc/hip/gemm_ops.hip)
# (a tuple so that the CPU unit tests of the operand checks can admit host tensors)
_DEVICE_TYPES: tuple[str, ...] = ("cuda",)


def _check(t: torch.Tensor, name: str) -> None:
    from nanofed_amd.core.exceptions import DeviceError

    if t.dtype != torch.bfloat16 or t.dim() != 2 or t.stride(1) != 1:
        raise DeviceError(f"gemm {name}: a bf16 matrix with unit column stride is required")
    if t.stride(0) % 8 or t.data_ptr() % 16:
        raise DeviceError(f"gemm {name}: row stride must be a multiple of 8 elements, base 16-byte aligned")


def _slab_buffer(dev: torch.device, floats: int) -> torch.Tensor:
    """fp32 scratch of a split-K launch.  Eager launches reuse one buffer per stream.
    A launch that is being captured into a graph gets a buffer of its own from the
    graph's memory pool: graphs captured one after the other on the same stream may
    later be replayed concurrently on different streams, and must not share scratch."""
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(floats, dtype=torch.float32, device=dev)
    key = (dev.index or 0, torch.cuda.current_stream(dev).cuda_stream)
    buf = _slabs.get(key)
    if buf is None or buf.numel() < floats:
        buf = torch.empty(floats, dtype=torch.float32, device=dev)
        _slabs[key] = buf
    return buf


GEO_128, GEO_256x128, GEO_256, GEO_128_2, GEO_128x192 = 0, 1, 2, 3, 4
_GEO_TILE = {GEO_128: (128, 128), GEO_256x128: (256, 128), GEO_256: (256, 256),
             GEO_128_2: (128, 128), GEO_128x192: (128, 192)}
# NANOFED_GEMM_GEO=0|1|2 pins the workgroup tile (measurements)
_FORCED_GEO = os.environ.get("NANOFED_GEMM_GEO")


def choose_geometry(M: int, N: int, K: int = 0, layout: int = 0, batch: int = 1) -> int:
    """Workgroup tile for an ``M x N`` output with reduction length ``K``.  Bigger
    tiles load fewer operand bytes per MFMA but make fewer workgroups and need a
    long K loop to pay for their prologue; the 128 x 128 tile with two LDS stages lets
    two workgroups share a compute unit, so one's prologue and epilogue overlap the
    other's main loop - the best choice at the GPT-2-small layer shapes in all three
    layouts.  The thresholds are read off ``tools/bench_gemm.py --geometries``.
    ``batch``: equally shaped problems that share the launch (their tiles add up)."""
    if _FORCED_GEO is not None:
        return int(_FORCED_GEO)
    ksteps = (K + 63) // 64
    t256 = (-(-M // 256)) * (-(-N // 256)) * batch
    # share of the last round of workgroups (one 256 x 256 tile per CU and round) that is used
    fill256 = (t256 / 256.0) / -(-t256 // 256)
    fits192 = layout == NT and N % 192 == 0
    if t256 >= 192 and min(M, N) >= 2048 and not (fits192 and (t256 < 256 or fill256 < 0.95)):
        # both sides wide and 256 x 256 tiles for three quarters of the CUs or more: the big
        # tile reads the fewest operand bytes per MFMA (4096 x 3072 x 768: 29.8 us against
        # 33.7 us; the LM head's forward 371 us against 442 us).  A k-contiguous output of
        # 192-column multiples is left to the 128 x 192 tile below when the big tiles would
        # leave part of a round empty (8 x 4096 x 2304: 4.5 rounds, 131.6 against 137.3 us).
        return GEO_256
    if t256 >= 48 and ksteps >= 256 and min(M, N) >= 512:
        # a very long reduction (the LM head's input gradient, K = 50257, cut along K):
        # 389 us against 472 us
        return GEO_256
    if layout == TN and t256 >= 192 and ksteps >= 32 and min(M, N) >= 512:
        # weight gradients of a group of clients: the big tile when its workgroups fill
        # whole rounds of the 256 CUs (8 x 2304 x 768: 216 tiles, one round, 122 us against
        # 152 us), not when a second round would be mostly empty (8 x 3072 x 768: 288
        # tiles, 246 us against 222 us)
        rounds = t256 / 256.0
        if rounds / -(-t256 // 256) >= 0.8:
            return GEO_256
    if layout == TN and t256 >= 224 and ksteps >= 32 and max(M, N) >= 16384 and min(M, N) >= 512:
        return GEO_256        # the LM head's weight gradient: one very long side (377 vs 408 us)
    if layout == NN and t256 >= 256 and ksteps >= 48 and min(M, N) >= 512:
        # data gradients of a group of clients with a long reduction (8 x 4096 x 768, K = 3072:
        # 177 against 188 us; at K = 2304 the small tile is still ahead, 128 against 133 us)
        return GEO_256
    if layout == TN and batch > 1 and t256 >= 256 and ksteps >= 32 and 512 <= M < N:
        # weight gradients of a group whose 256 x 256 tiles would leave a round mostly empty,
        # with the short side first (8 x 768 x 3072: 232 against 250 us; 8 x 3072 x 768 stays
        # on the small tile, 219 against 224 us)
        return GEO_256x128
    if t256 >= 1024 and layout == NT and not fits192:
        return GEO_256x128    # very wide outputs with a short K loop (the LM head)
    if layout == NT and N % 192 == 0:
        # k-contiguous operands may use 128 x 192 tiles: when they make about one full
        # round of workgroups (two per CU) where 128 x 128 tiles make one and a half
        t192 = (-(-M // 128)) * (N // 192) * batch
        t128 = (-(-M // 128)) * (-(-N // 128)) * batch
        if 256 <= t192 <= 512 < t128 or t192 >= 768:
            # (many tiles - the launches of a group of clients: 8 x 4096 x 768 x 768 46.2
            # against 50.6 us, x 3072 deep 156.4 against 177.7 us)
            return GEO_128x192
    return GEO_128_2


def _launch(layout: int, a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, M: int, N: int,
            K: int, bias: torch.Tensor | None = None, out2: torch.Tensor | None = None,
            splits: int | None = None, geo: int | None = None) -> None:
    from ctypes import c_int, c_long, c_void_p

    from nanofed_amd.core.exceptions import DeviceError

    for t, name in ((a, "A"), (b, "B"), (out, "C")):
        _check(t, name)
    if out2 is not None:
        _check(out2, "C2")
        if out2.stride(0) != out.stride(0) or out2.shape != out.shape:
            raise DeviceError("gemm: the second output must have the shape and stride of the first")
    if bias is not None and (bias.dtype != torch.bfloat16 or bias.numel() != N
                             or not bias.is_contiguous()):
        raise DeviceError("gemm: bias must be a contiguous bf16 vector of N entries")
    if tuple(out.shape) != (M, N):
        raise DeviceError(f"gemm: output is {tuple(out.shape)}, expected {(M, N)}")
    # operand shapes per layout: a slip here would be an out-of-bounds device access
    want_a, want_b = {NT: ((M, K), (N, K)), NN: ((M, K), (K, N)), TN: ((K, M), (K, N))}[layout]
    if tuple(a.shape) != want_a or tuple(b.shape) != want_b:
        raise DeviceError(
            f"gemm: operands are {tuple(a.shape)} and {tuple(b.shape)}, layout {layout} with "
            f"M, N, K = {M}, {N}, {K} needs {want_a} and {want_b}")
    if M < 1 or N < 1 or K < 1:
        raise DeviceError("gemm: empty problem")
    for t in (a, b, out2, bias):
        if t is not None and t.device != out.device:
            raise DeviceError("gemm: operands live on different devices")
    if out.device.type not in _DEVICE_TYPES:
        raise DeviceError("gemm: device tensors are required")
    # the epilogue stores whole 16-byte pieces: columns N .. round8(N) - 1 of every row are
    # written too, so they must belong to this matrix's rows (padding), not to a neighbour
    n8 = (N + 7) // 8 * 8
    for t in (out, out2):
        if t is None or n8 == N or t.shape[0] == 0:
            continue
        col0 = t.storage_offset() % t.stride(0) if t.stride(0) else 0
        if t.stride(0) < n8 or col0 != 0:
            raise DeviceError(
                f"gemm: N = {N} is not a multiple of 8: the output must start at column 0 of "
                f"rows at least {n8} long (the columns up to {n8} are written as padding), "
                "not be a column slice inside a wider matrix")
    lib = library.hip_lib()
    fn = lib.nf_gemm_bf16
    if not getattr(fn, "_declared", False):
        fn.argtypes = [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                       c_int, c_int, c_int, c_long, c_long, c_long, c_int, c_void_p]
        fn.restype = c_int
        fn._declared = True
    if geo is None:
        geo = choose_geometry(M, N, K, layout)
    if splits is None:
        splits = choose_splits(M, N, K, geo)
    slabs = _slab_buffer(out.device, splits * M * N) if splits > 1 else None
    library.check(
        fn(layout, geo, a.data_ptr(), b.data_ptr(), out.data_ptr(),
           out2.data_ptr() if out2 is not None else None,
           bias.data_ptr() if bias is not None else None,
           slabs.data_ptr() if slabs is not None else None,
           M, N, K, a.stride(0), b.stride(0), out.stride(0), splits,
           library.stream_ptr(out.device)),
        "nf_gemm_bf16",
    )


def linear_fwd(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None, out: torch.Tensor,
               gelu_out: torch.Tensor | None = None) -> None:
    """``out = x @ w.T (+ bias)``; with ``gelu_out`` also ``gelu_out = gelu(out)``
    (tanh form, applied to the bf16-rounded ``out``).  x ``[M, K]``, w ``[N, K]``."""
    M, K = x.shape
    N = w.shape[0]
    if x.device.type == "cuda" and enabled(x.device):
        _launch(NT, x, w, out, M, N, K, bias=bias, out2=gelu_out)
        return
    y = x.float() @ w.float().t()
    if bias is not None:
        y = y + bias.float()
    out.copy_(y.to(torch.bfloat16))
    if gelu_out is not None:
        gelu_out.copy_(torch.nn.functional.gelu(out.float(), approximate="tanh").to(torch.bfloat16))


def linear_dgrad(dy: torch.Tensor, w: torch.Tensor, out: torch.Tensor) -> None:
    """``out = dy @ w``: dy ``[M, N]``, w ``[N, K]`` (as stored), out ``[M, K]``."""
    M, N = dy.shape
    K = w.shape[1]
    if dy.device.type == "cuda" and enabled(dy.device):
        _launch(NN, dy, w, out, M, K, N)
        return
    out.copy_((dy.float() @ w.float()).to(torch.bfloat16))


def choose_splits(M: int, N: int, K: int, geo: int = GEO_128, batch: int = 1) -> int:
    """Split factor of a GEMM with ``M x N`` outputs and reduction length ``K``.  A
    launch with fewer tiles than the chip has workgroup slots is cut along K until tiles
    times ranges come to about 430 workgroups (two per CU, a little under two full
    rounds - measured optimum at the GPT-2-small weight-gradient shapes), provided every
    range keeps at least 8 K steps of 64.  Outputs of a few tiles only (weight gradients
    of small layers: K is the number of pixels or tokens) have almost no slab traffic:
    they are cut into ranges as short as 4 steps, up to 64."""
    forced = os.environ.get("NANOFED_GEMM_SPLITS")
    if forced is not None:
        return int(forced)
    bm, bn = _GEO_TILE[geo]
    tiles = (-(-M // bm)) * (-(-N // bn)) * batch
    ksteps = (K + 63) // 64
    if tiles >= 200:
        if geo == GEO_256 and tiles < 1024 and ksteps >= 512:
            # one 256 x 256 workgroup per CU: a launch of a round and a half takes two rounds.
            # With a very long reduction the slabs cost little next to the empty half round
            # (the LM head's input gradient for 8 clients, 384 tiles, K = 50257: 2987 us
            # unsplit, 2408 us cut in two, 2476 in four; at K = 3072 the slabs cost more than
            # the half round: 182 against 228 us)
            def fill(t: int) -> float:
                return (t / 256.0) / -(-t // 256)
            if fill(tiles) < 0.8:
                for s in (2, 3, 4):
                    if fill(tiles * s) >= 0.95:
                        return s
        return 1
    if tiles <= 16:
        return max(1, min(-(-256 // tiles), 64, ksteps // 4))
    want = max(1, (432 + tiles // 2) // tiles)
    return max(1, min(want, 16, ksteps // 8))


def linear_wgrad(dy: torch.Tensor, x: torch.Tensor, out: torch.Tensor) -> None:
    """``out = dy.T @ x``: dy ``[R, N]``, x ``[R, K]``, out ``[N, K]`` - the
    reduction runs over the rows, both operands are read as they are stored."""
    R, N = dy.shape
    K = x.shape[1]
    if dy.device.type == "cuda" and enabled(dy.device):
        _launch(TN, dy, x, out, N, K, R)
        return
    out.copy_((dy.float().t() @ x.float()).to(torch.bfloat16))


def symbols() -> list[str]:
    return ["nf_gemm_bf16", "nf_gemm_geometries"]
