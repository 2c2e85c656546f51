"""Tile rows walked together by the workgroups of an XCD (NANOFED_GEMM_GROUPM): time of the
batched forward / data-gradient GEMMs of the GPT-2 step with 256 x 256 tiles."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nanofed_amd.ops import group_ops as go  # noqa: E402
from tools.bench_gemm import timed  # noqa: E402

torch.cuda.set_device(0)
bf = torch.bfloat16
M, G = 4096, 8
for name, (N, K) in {"c_fc": (3072, 768), "c_attn": (2304, 768), "mlp_proj": (768, 3072), "lm_head": (50257, 768)}.items():
    x = torch.randn(G, M, K, device="cuda").to(bf)
    w = (torch.randn(G, N, K, device="cuda") * 0.05).to(bf)
    LD = (N + 63) // 64 * 64
    y = torch.empty(G, M, LD, dtype=bf, device="cuda")
    dy = (torch.randn(G, M, LD, device="cuda") * 0.1).to(bf)
    dx = torch.empty(G, M, K, dtype=bf, device="cuda")
    dw = torch.empty(G, N, K, dtype=bf, device="cuda")
    slabs = torch.empty(max(go.gemm_slab_floats(M, N, K, 0, G), go.gemm_slab_floats(M, K, N, 1, G),
                            go.gemm_slab_floats(N, K, M, 2, G), 1), device="cuda")
    fns = {
        "fwd": (lambda: go.gemm_batched(0, x, w, y, M, N, K, K, K, LD, G, M * K, N * K, M * LD, slabs, go._stream(y)), y),
        "dgrad": (lambda: go.gemm_batched(1, dy, w, dx, M, K, N, LD, K, K, G, M * LD, N * K, M * K, slabs, go._stream(y)), dx),
        "wgrad": (lambda: go.gemm_batched(2, dy, x, dw, N, K, M, LD, K, K, G, M * LD, M * K, N * K, slabs, go._stream(y)), dw),
    }
    for tag, (fn, out) in fns.items():
        row, ref = [], None
        for gm in (4, 1, 8, 2, 1, 4):
            os.environ["NANOFED_GEMM_GROUPM"] = str(gm)
            fn()
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            same = torch.equal(ref, out)
            row.append("%d: %.1f%s" % (gm, timed(fn, 3, 5), "" if same else " DIFFERENT"))
        print("%-9s %-6s %s" % (name, tag, "  ".join(row)), flush=True)
    del x, w, y, dy, dx, dw, slabs
