"""Microseconds of the batched GEMMs of a grouped GPT-2-small step (8 clients x 4096 rows)
for every workgroup tile - the table behind the ``batch`` rules of ``choose_geometry``.

    python tools/probes/gemm_batched_sweep.py [batch]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nanofed_amd.ops import gemm_ops  # noqa: E402
from nanofed_amd.ops import group_ops as go  # noqa: E402
from tools.bench_gemm import timed  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
R = 4096
bf = torch.bfloat16
torch.cuda.set_device(0)
slabs = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
shapes = {"c_attn": (2304, 768), "attn_proj": (768, 768), "c_fc": (3072, 768), "mlp_proj": (768, 3072),
          "lm_head": (50257, 768)}
if len(sys.argv) > 2:
    shapes = {k: v for k, v in shapes.items() if k in sys.argv[2:]}
for name, (N, K) in shapes.items():
    ld = (N + 63) // 64 * 64
    x = torch.randn(G * R, K, device="cuda").to(bf)
    w = (torch.randn(G, N, K, device="cuda") * 0.05).to(bf)
    y = torch.empty(G * R, ld, dtype=bf, device="cuda")
    dy = torch.empty(G * R, ld, dtype=bf, device="cuda").normal_(0, 0.1)
    dx = torch.empty(G * R, K, dtype=bf, device="cuda")
    dw = torch.empty(G, N, K, dtype=bf, device="cuda")
    kinds = {
        "fwd": (lambda: go.gemm_batched(gemm_ops.NT, x, w, y, R, N, K, K, K, ld, G, R * K, N * K, R * ld, slabs, go._stream(x)),
                (R, N, K, gemm_ops.NT)),
        "dgrad": (lambda: go.gemm_batched(gemm_ops.NN, dy, w, dx, R, K, N, ld, K, K, G, R * ld, N * K, R * K, slabs, go._stream(x)),
                  (R, K, N, gemm_ops.NN)),
        "wgrad": (lambda: go.gemm_batched(gemm_ops.TN, dy, x, dw, N, K, R, ld, K, K, G, R * ld, R * K, N * K, slabs, go._stream(x)),
                  (N, K, R, gemm_ops.TN)),
    }
    for kind, (fn, (m, n, k, lay)) in kinds.items():
        gemm_ops._FORCED_GEO = None
        chosen = gemm_ops.choose_geometry(m, n, k, lay, G)
        row = []
        for geo in (0, 1, 2, 3, 4):
            if geo == 4 and (lay != gemm_ops.NT or n % 192):
                continue
            gemm_ops._FORCED_GEO = str(geo)
            row.append(f"g{geo}={timed(fn, launches=4, reps=5):.1f}")
        gemm_ops._FORCED_GEO = None
        flops = 2.0 * G * m * n * k
        best = min(float(v.split("=")[1]) for v in row)
        print(f"{name} {kind} M{m} N{n} K{k}: chosen g{chosen} " + " ".join(row) +
              f"  best {flops / best / 1e6:.0f} TFLOP/s", flush=True)
