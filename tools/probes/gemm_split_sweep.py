"""Microseconds of one GEMM shape for every workgroup tile and split-K factor - the table
``choose_geometry`` / ``choose_splits`` in ``nanofed_amd/ops/gemm_ops.py`` are read off.

    python tools/probes/gemm_split_sweep.py M N K [layout: 0 NT | 1 NN | 2 TN]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nanofed_amd.ops import gemm_ops  # noqa: E402
from tools.bench_gemm import timed  # noqa: E402

M, N, K = (int(v) for v in sys.argv[1:4])
layout = int(sys.argv[4]) if len(sys.argv) > 4 else 0
bf = torch.bfloat16
torch.cuda.set_device(0)
shape_a, shape_b = {0: ((M, K), (N, K)), 1: ((M, K), (K, N)), 2: ((K, M), (K, N))}[layout]
a = torch.randn(*shape_a, device="cuda").to(bf)
b = (torch.randn(*shape_b, device="cuda") * 0.05).to(bf)
out = torch.empty(M, N, dtype=bf, device="cuda")
lib = {0: lambda: torch.mm(a, b.t(), out=out), 1: lambda: torch.mm(a, b, out=out),
       2: lambda: torch.mm(a.t(), b, out=out)}[layout]
print(f"M N K = {M} {N} {K} layout {layout}: library {timed(lib):.2f} us", flush=True)
for geo in (0, 1, 2, 3, 4):
    if geo == 4 and (layout != 0 or N % 192):
        continue
    row = []
    for splits in (1, 2, 3, 4, 6, 8):
        if (K + 63) // 64 // splits < 4:
            break
        t = timed(lambda: gemm_ops._launch(layout, a, b, out, M, N, K, splits=splits, geo=geo))
        row.append(f"s{splits}={t:.2f}")
    print(f"geo {geo}: " + " ".join(row), flush=True)
