import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nanofed_amd.ops import gemm_ops
from tools.bench_gemm import timed
torch.cuda.set_device(0)
bf = torch.bfloat16
for (M, N, K) in [(4096, 4096, 4096), (8192, 8192, 8192), (4096, 2304, 768), (4096, 2304, 3072), (4096, 2304, 12288)]:
    x = torch.randn(M, K, device="cuda").to(bf)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(bf)
    y = torch.empty(M, N, dtype=bf, device="cuda")
    dy = (torch.randn(M, N, device="cuda") * 0.1).to(bf)
    dx = torch.empty(M, K, dtype=bf, device="cuda")
    dw = torch.empty(N, K, dtype=bf, device="cuda")
    fl = 2.0 * M * N * K
    t1 = timed(lambda: gemm_ops.linear_fwd(x, w, None, y), 3, 5)
    t2 = timed(lambda: torch.mm(x, w.t(), out=y), 3, 5)
    t3 = timed(lambda: gemm_ops.linear_dgrad(dy, w, dx), 3, 5)
    t4 = timed(lambda: gemm_ops.linear_wgrad(dy, x, dw), 3, 5)
    print(M, N, K, "fwd %.1f us %.0f TF | lib %.1f us %.0f TF | dgrad %.0f TF | wgrad %.0f TF" % (t1, fl/t1/1e6, t2, fl/t2/1e6, fl/t3/1e6, fl/t4/1e6), flush=True)
