"""Waves 4-7 one MFMA segment behind waves 0-3 (NANOFED_GEMM_STAGGER=1) against the plain
loop: microseconds for the batched GPT-2 layer GEMMs (8 clients), the LM head and two
squares, and whether the outputs are the same bits.  The staggered loop was measured with
this probe and not kept (result in profiles/r3_gemm_stagger_ab.txt):
the kernel in the tree ignores the variable, so both columns now time the same code."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nanofed_amd.ops import gemm_ops  # noqa: E402
from nanofed_amd.ops import group_ops as go  # noqa: E402
from tools.bench_gemm import timed  # noqa: E402

torch.cuda.set_device(0)
bf = torch.bfloat16


def ab(name, fn, out):
    res, outs = [], []
    for mode in ("0", "1"):
        os.environ["NANOFED_GEMM_STAGGER"] = mode
        out.zero_()
        fn()
        torch.cuda.synchronize()
        outs.append(out.clone())
        res.append(timed(fn, 3, 5))
    os.environ["NANOFED_GEMM_STAGGER"] = "0"
    print("%-22s plain %8.1f us  staggered %8.1f us  (%+.1f %%)  same bits: %s" % (
        name, res[0], res[1], 100.0 * (res[1] / res[0] - 1.0), torch.equal(outs[0], outs[1])), flush=True)
    return res


M, G = 4096, 8
tot = [0.0, 0.0]
shapes = {"c_attn": (2304, 768), "attn_proj": (768, 768), "c_fc": (3072, 768), "mlp_proj": (768, 3072),
          "lm_head": (50257, 768)}
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
    for tag, fn, out in (
        ("fwd", lambda: go.gemm_batched(0, x, w, y, M, N, K, K, K, LD, G, M * K, N * K, M * LD, slabs, go._stream(y)), y),
        ("dgrad", lambda: go.gemm_batched(1, dy, w, dx, M, K, N, LD, K, K, G, M * LD, N * K, M * K, slabs, go._stream(y)), dx),
        ("wgrad", lambda: go.gemm_batched(2, dy, x, dw, N, K, M, LD, K, K, G, M * LD, M * K, N * K, slabs, go._stream(y)), dw),
    ):
        r = ab(name + " " + tag + " x8", fn, out)
        mult = 1 if name == "lm_head" else 12
        tot[0] += mult * r[0]
        tot[1] += mult * r[1]
    del x, w, y, dy, dx, dw, slabs
print("12 layers + LM head, 8 clients: plain %.2f ms, staggered %.2f ms" % (tot[0] / 1e3, tot[1] / 1e3))

for n in (4096, 8192):
    a = torch.randn(n, n, device="cuda").to(bf)
    b = (torch.randn(n, n, device="cuda") * 0.05).to(bf)
    c = torch.empty(n, n, dtype=bf, device="cuda")
    r = ab("square %d NT" % n, lambda: gemm_ops.linear_fwd(a, b, None, c), c)
    print("   %.0f / %.0f TFLOP/s" % (2.0 * n ** 3 / r[0] / 1e6, 2.0 * n ** 3 / r[1] / 1e6))
    r = ab("square %d NN" % n, lambda: gemm_ops.linear_dgrad(a, b, c), c)
    r = ab("square %d TN" % n, lambda: gemm_ops.linear_wgrad(a, b, c), c)
