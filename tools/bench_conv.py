"""Microseconds per 3x3 convolution pass (forward, input gradient, weight
gradient): the hand-written kernels against the library convolution, for the
four ResNet-18 stage shapes at batch 64 (or --batch).

    python tools/bench_conv.py [--reps 50] [--min-wgs 192] [--batch 512]
"""

from __future__ import annotations

import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CL = torch.channels_last
SHAPES = [(64, 32, 32, 64, 64), (64, 16, 16, 128, 128), (64, 8, 8, 256, 256), (64, 4, 4, 512, 512)]


def timed(fn, reps: int) -> float:
    for _ in range(5):
        fn()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(graph):
        for _ in range(reps):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    graph.replay()
    t1.record()
    torch.cuda.synchronize()
    return 1000.0 * t0.elapsed_time(t1) / reps


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--min-wgs", type=int, default=None)
    ap.add_argument("--no-library", action="store_true")
    ap.add_argument("--batch", type=int, default=64,
                    help="images per launch (512 = the eight clients of the grouped ResNet trainer)")
    args = ap.parse_args()
    from nanofed_amd.ops import conv_ops

    if args.min_wgs is not None:
        conv_ops.set_wgrad_min_workgroups(args.min_wgs)
    torch.backends.cudnn.benchmark = True
    out = {}
    for b, h, w, ci, co in SHAPES:
        b = args.batch
        x = torch.randn(b, ci, h, w, device="cuda").to(torch.bfloat16).contiguous(memory_format=CL)
        wt = (torch.randn(co, ci, 3, 3, device="cuda") * 0.02).to(torch.bfloat16)
        wt = wt.contiguous(memory_format=CL)
        dy = torch.randn(b, co, h, w, device="cuda").to(torch.bfloat16).contiguous(memory_format=CL)
        flops = 2.0 * b * h * w * ci * co * 9
        row = {
            "fwd": timed(lambda: conv_ops.conv3x3_forward(x, wt), args.reps),
            "dgrad": timed(lambda: conv_ops.conv3x3_input_grad(dy, wt), args.reps),
            "wgrad": timed(lambda: conv_ops.conv3x3_weight_grad(x, dy), args.reps),
        }
        if not args.no_library:
            row["lib_fwd"] = timed(lambda: torch.nn.functional.conv2d(x, wt, padding=1), args.reps)
            grads = lambda mask: torch.ops.aten.convolution_backward(  # noqa: E731
                dy, x, wt, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, mask)
            row["lib_dgrad"] = timed(lambda: grads([True, False, False]), args.reps)
            row["lib_wgrad"] = timed(lambda: grads([False, True, False]), args.reps)
        row = {k: round(v, 2) for k, v in row.items()}
        row["tflops_fwd"] = round(flops / row["fwd"] / 1e6, 1)
        row["tflops_wgrad"] = round(flops / row["wgrad"] / 1e6, 1)
        out[f"{b}x{h}x{w}x{ci}->{co}"] = row
        print(f"{b}x{h}x{w}x{ci}->{co}", json.dumps(row), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
