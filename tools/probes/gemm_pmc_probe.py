"""A few launches of every GEMM geometry and layout at 4096^3 (for hardware-counter passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nanofed_amd.ops import gemm_ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
bf = torch.bfloat16
a = torch.randn(n, n, device="cuda").to(bf)
b = (torch.randn(n, n, device="cuda") * 0.05).to(bf)
c = torch.empty(n, n, device="cuda", dtype=bf)
for geo in (0, 1, 2):
    for layout in (gemm_ops.NT, gemm_ops.NN, gemm_ops.TN):
        for _ in range(3):
            gemm_ops._launch(layout, a, b, c, n, n, n, geo=geo, splits=1)
torch.cuda.synchronize()
