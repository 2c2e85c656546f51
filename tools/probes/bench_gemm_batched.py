"""GPT-2-small layer GEMMs for 8 clients: eight single launches (what the per-client
trainer issues) against one batched launch (per-client weights), microseconds per client."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nanofed_amd.ops import gemm_ops  # noqa: E402
from nanofed_amd.ops import group_ops as go  # noqa: E402
from tools.bench_gemm import timed  # noqa: E402

torch.cuda.set_device(0)
bf = torch.bfloat16
M, G = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 8
shapes = {"c_attn": (2304, 768), "attn_proj": (768, 768), "c_fc": (3072, 768), "mlp_proj": (768, 3072)}
if len(sys.argv) > 2 and sys.argv[2] == "lm_head":
    shapes = {"lm_head": (50257, 768)}
tot_single = tot_batched = 0.0
for name, (N, K) in shapes.items():
    x = torch.randn(G, M, K, device="cuda").to(bf)
    w = (torch.randn(G, N, K, device="cuda") * 0.05).to(bf)
    LD = (N + 63) // 64 * 64
    y = torch.empty(G, M, LD, dtype=bf, device="cuda")
    dy = (torch.randn(G, M, LD, device="cuda") * 0.1).to(bf)
    dx = torch.empty(G, M, K, dtype=bf, device="cuda")
    dw = torch.empty(G, N, K, dtype=bf, device="cuda")
    slabs = torch.empty(max(go.gemm_slab_floats(M, N, K, 0, G), go.gemm_slab_floats(M, K, N, 1, G),
                            go.gemm_slab_floats(N, K, M, 2, G), 1), device="cuda")

    def single():
        for g in range(G):
            gemm_ops.linear_fwd(x[g], w[g], None, y[g][:, :N])

    def single_d():
        for g in range(G):
            gemm_ops.linear_dgrad(dy[g][:, :N], w[g], dx[g])

    def single_w():
        for g in range(G):
            gemm_ops.linear_wgrad(dy[g][:, :N], x[g], dw[g])

    def batched():
        go.gemm_batched(0, x, w, y, M, N, K, K, K, LD, G,
                        M * K, N * K, M * LD, slabs, go._stream(y))

    def batched_d():
        go.gemm_batched(1, dy, w, dx, M, K, N, LD, K, K, G,
                        M * LD, N * K, M * K, slabs, go._stream(y))

    def batched_w():
        go.gemm_batched(2, dy, x, dw, N, K, M, LD, K, K, G,
                        M * LD, M * K, N * K, slabs, go._stream(y))

    row = {}
    for tag, a, b in (("fwd", single, batched), ("dgrad", single_d, batched_d), ("wgrad", single_w, batched_w)):
        ta, tb = timed(a, 3, 5) / G, timed(b, 3, 5) / G
        tot_single += ta
        tot_batched += tb
        row[tag] = (round(ta, 1), round(tb, 1))
    print(name, row, flush=True)
print("per layer and client: single %.1f us, batched %.1f us" % (tot_single, tot_batched))
