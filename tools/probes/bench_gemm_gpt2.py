import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nanofed_amd.ops import gemm_ops
from tools.bench_gemm import timed
torch.cuda.set_device(0)
bf = torch.bfloat16
M = 4096
geos = [int(g) for g in (sys.argv[1] if len(sys.argv) > 1 else "0,1,2,3").split(",")]
for name, (N, K) in {"c_attn": (2304, 768), "attn_proj": (768, 768), "c_fc": (3072, 768), "mlp_proj": (768, 3072)}.items():
    x = torch.randn(M, K, device="cuda").to(bf); w = (torch.randn(N, K, device="cuda") * 0.05).to(bf)
    y = torch.empty(M, N, dtype=bf, device="cuda"); dy = (torch.randn(M, N, device="cuda") * 0.1).to(bf)
    dx = torch.empty(M, K, dtype=bf, device="cuda"); dw = torch.empty(N, K, dtype=bf, device="cuda")
    lib = [timed(lambda: torch.mm(x, w.t(), out=y), 5, 5), timed(lambda: torch.mm(dy, w, out=dx), 5, 5),
           timed(lambda: torch.mm(dy.t(), x, out=dw), 5, 5)]
    row = {"lib": [round(v, 1) for v in lib]}
    for g in geos:
        f = timed(lambda: gemm_ops._launch(gemm_ops.NT, x, w, y, M, N, K, geo=g), 5, 5)
        if g == 4:
            row[g] = [round(f, 1)]
            continue
        d = timed(lambda: gemm_ops._launch(gemm_ops.NN, dy, w, dx, M, K, N, geo=g), 5, 5)
        sp = gemm_ops.choose_splits(N, K, M, g)
        wg = timed(lambda: gemm_ops._launch(gemm_ops.TN, dy, x, dw, N, K, M, geo=g, splits=sp), 5, 5)
        row[g] = [round(f, 1), round(d, 1), round(wg, 1)]
    print(name, row, flush=True)
