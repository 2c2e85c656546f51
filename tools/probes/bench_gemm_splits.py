import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nanofed_amd.ops import gemm_ops
from tools.bench_gemm import timed
torch.cuda.set_device(0)
bf = torch.bfloat16
M = 4096
for name, (N, K) in {"c_attn": (2304, 768), "attn_proj": (768, 768), "c_fc": (3072, 768), "mlp_proj": (768, 3072)}.items():
    x = torch.randn(M, K, device="cuda").to(bf); w = (torch.randn(N, K, device="cuda") * 0.05).to(bf)
    y = torch.empty(M, N, dtype=bf, device="cuda"); dy = (torch.randn(M, N, device="cuda") * 0.1).to(bf)
    dx = torch.empty(M, K, dtype=bf, device="cuda"); dw = torch.empty(N, K, dtype=bf, device="cuda")
    row = {}
    for kind, fn in (("fwd", lambda s: gemm_ops._launch(gemm_ops.NT, x, w, y, M, N, K, geo=3, splits=s)),
                     ("dgrad", lambda s: gemm_ops._launch(gemm_ops.NN, dy, w, dx, M, K, N, geo=3, splits=s)),
                     ("wgrad", lambda s: gemm_ops._launch(gemm_ops.TN, dy, x, dw, N, K, M, geo=3, splits=s))):
        row[kind] = {s: round(timed(lambda: fn(s), 5, 5), 1) for s in (1, 2, 3, 4, 6, 8)}
    print(name, row, flush=True)
