"""Split factors for the batched data-gradient launches whose 256 x 256 tiles leave the last
round of workgroups half empty (8 clients x 4096 x 768: 384 tiles on 256 CUs)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nanofed_amd.ops import group_ops as go  # noqa: E402
from tools.bench_gemm import timed  # noqa: E402

torch.cuda.set_device(0)
bf = torch.bfloat16
M, G = 4096, 8
for name, (N, K) in {"lm_head": (50257, 768), "c_fc": (3072, 768), "c_attn": (2304, 768)}.items():
    w = (torch.randn(G, N, K, device="cuda") * 0.05).to(bf)
    LD = (N + 63) // 64 * 64
    dy = (torch.randn(G, M, LD, device="cuda") * 0.1).to(bf)
    dx = torch.empty(G, M, K, dtype=bf, device="cuda")
    slabs = torch.empty(G * 4 * M * K, device="cuda")
    ref = None
    for geo in ("2", "3"):
        os.environ["NANOFED_GEMM_GEO"] = geo
        import nanofed_amd.ops.gemm_ops as gm
        gm._FORCED_GEO = geo
        for sp in (1, 2, 3, 4):
            os.environ["NANOFED_GEMM_SPLITS"] = str(sp)
            fn = lambda: go.gemm_batched(1, dy, w, dx, M, K, N, LD, K, K, G, M * LD, N * K, M * K, slabs, go._stream(dx))
            fn()
            torch.cuda.synchronize()
            if ref is None:
                ref = dx.float().clone()
            err = (dx.float() - ref).abs().max().item() / ref.abs().max().item()
            print("%-8s dgrad x8 geo %s splits %d: %8.1f us   (max diff to the first %.2e)" % (name, geo, sp, timed(fn, 3, 5), err), flush=True)
