"""Microseconds per pass of the ResNet-18 convolutions that run as patch gather + GEMM,
against the autotuned library (batch 64, timed inside hipGraphs)."""
import json
import sys

import torch

from nanofed_amd.ops import conv_ops

CL = torch.channels_last
SHAPES = {"s2_64_128": (64, 128, 3, 2, 1, 32), "s2_128_256": (128, 256, 3, 2, 1, 16),
          "s2_256_512": (256, 512, 3, 2, 1, 8), "sc_64_128": (64, 128, 1, 2, 0, 32),
          "sc_128_256": (128, 256, 1, 2, 0, 16), "sc_256_512": (256, 512, 1, 2, 0, 8),
          "stem": (3, 64, 3, 1, 1, 32), "s1_512_4x4": (512, 512, 3, 1, 1, 4)}


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(5):
        g.replay()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) * 1000 / (5 * iters)


torch.backends.cudnn.benchmark = True
out = {}
for name, (cin, cout, k, s, p, hw) in SHAPES.items():
    x = torch.randn(64, cin, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=CL)
    w = torch.randn(cout, cin, k, k, device="cuda").to(torch.bfloat16).contiguous(memory_format=CL)
    ho = (hw + 2 * p - k) // s + 1
    dy = torch.randn(64, cout, ho, ho, device="cuda").to(torch.bfloat16).contiguous(memory_format=CL)
    col = conv_ops.im2col(x, k, k, s, p)
    w2d = conv_ops._weight2d(w, col.shape[1])
    from nanofed_amd.ops import gemm_ops
    y2d = torch.empty(col.shape[0], cout, dtype=torch.bfloat16, device="cuda")
    dw2d = torch.empty_like(w2d)
    dcol = torch.empty_like(col)
    dy2d = dy.permute(0, 2, 3, 1).reshape(-1, cout)
    row = {
        "im2col": timed(lambda: conv_ops.im2col(x, k, k, s, p)),
        "fwd_gemm": timed(lambda: gemm_ops.linear_fwd(col, w2d, None, y2d)),
        "wgrad_gemm": timed(lambda: gemm_ops.linear_wgrad(dy2d, col, dw2d)),
        "lib_fwd": timed(lambda: torch.ops.aten.convolution(x, w, None, [s, s], [p, p], [1, 1], False, [0, 0], 1)),
        "lib_wgrad": timed(lambda: torch.ops.aten.convolution_backward(
            dy, x, w, None, [s, s], [p, p], [1, 1], False, [0, 0], 1, [False, True, False])),
    }
    if cin % 8 == 0:
        row["dgrad_gemm"] = timed(lambda: gemm_ops.linear_dgrad(dy2d, w2d, dcol))
        row["col2im"] = timed(lambda: conv_ops.col2im(dcol, tuple(x.shape), k, k, s, p))
        row["lib_dgrad"] = timed(lambda: torch.ops.aten.convolution_backward(
            dy, x, w, None, [s, s], [p, p], [1, 1], False, [0, 0], 1, [True, False, False]))
    out[name] = {k2: round(v, 2) for k2, v in row.items()}
    print(name, out[name], flush=True)
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "/dev/stdout", "w"), indent=1)
