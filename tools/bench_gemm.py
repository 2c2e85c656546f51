"""Microseconds per GEMM of the GPT-2-small step (4096 rows) for the hand-written
MFMA kernels next to the vendor library call they replace (``torch.mm`` /
``torch.addmm`` = hipBLASLt), on the same random data, timed inside hipGraphs.

    python tools/bench_gemm.py [--rows 4096] [--out profiles/gemm_us.json]
"""

from __future__ import annotations

import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from nanofed_amd.ops import gemm_ops  # noqa: E402


def timed(fn, launches=10, reps=10):
    fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(launches):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1000.0 / (reps * launches))
    return best


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=4096)
    ap.add_argument("--out", default="")
    ap.add_argument("--lm-head", action="store_true", help="include the 50257-column LM head")
    ap.add_argument("--geometries", action="store_true", help="also time every workgroup tile")
    args = ap.parse_args()
    torch.cuda.set_device(0)
    M = args.rows
    shapes = {"c_attn": (2304, 768), "attn_proj": (768, 768), "c_fc": (3072, 768),
              "mlp_proj": (768, 3072)}
    if args.lm_head:
        shapes["lm_head"] = (50257, 768)
    bf = torch.bfloat16
    out = {}
    for name, (N, K) in shapes.items():
        x = torch.randn(M, K, device="cuda").to(bf)
        w = (torch.randn(N, K, device="cuda") * 0.05).to(bf)
        b = torch.randn(N, device="cuda").to(bf)
        ld = (N + 63) // 64 * 64
        ybuf = torch.empty(M, ld, dtype=bf, device="cuda")
        y = ybuf[:, :N]
        dy = (torch.randn(M, ld, device="cuda") * 0.1).to(bf)[:, :N]
        dx = torch.empty(M, K, dtype=bf, device="cuda")
        dw = torch.empty(N, K, dtype=bf, device="cuda")
        flops = 2.0 * M * N * K
        row = {}
        for kind, mine, lib in (
            ("fwd", lambda: gemm_ops.linear_fwd(x, w, b, y), lambda: torch.addmm(b, x, w.t(), out=y)),
            ("dgrad", lambda: gemm_ops.linear_dgrad(dy, w, dx), lambda: torch.mm(dy, w, out=dx)),
            ("wgrad", lambda: gemm_ops.linear_wgrad(dy, x, dw), lambda: torch.mm(dy.t(), x, out=dw)),
        ):
            t_mine, t_lib = timed(mine), timed(lib)
            row[kind] = {"us": round(t_mine, 2), "lib_us": round(t_lib, 2),
                         "tflops": round(flops / t_mine / 1e6, 1),
                         "lib_tflops": round(flops / t_lib / 1e6, 1)}
            if args.geometries:
                per_geo = {}
                for geo in (0, 1, 2, 3):
                    gemm_ops._FORCED_GEO = str(geo)
                    per_geo[geo] = round(timed(mine), 2)
                gemm_ops._FORCED_GEO = None
                row[kind]["us_by_geometry"] = per_geo
        out[name] = row
        print(name, json.dumps(row), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump({"rows": M, "shapes": out}, f, indent=1)


if __name__ == "__main__":
    main()
