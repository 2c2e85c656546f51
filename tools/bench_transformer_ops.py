"""Microseconds per launch of the transformer kernels at GPT-2-small shapes
(4 x 1024 tokens, 12 heads), timed inside hipGraphs, next to the PyTorch ops
they replace.    python tools/bench_transformer_ops.py
"""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nanofed_amd.ops import attention as attn_ops  # noqa: E402
from nanofed_amd.ops import nn_ops  # noqa: E402

B, T, H = 4, 1024, 12
C, R, V = H * 64, B * T, 50257
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
bf = torch.bfloat16


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).to(bf)


def timed(fn, launches=20, reps=5):
    fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(launches):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) * 1000.0 / (reps * launches), 2)


qkv, dout = rnd(R, 3 * C), rnd(R, C)
out, dqkv = torch.empty(R, C, dtype=bf, device=dev), torch.empty(R, 3 * C, dtype=bf, device=dev)
lse = torch.empty(B, H, T, device=dev)
scratch = attn_ops.backward_scratch(B, H, T, dev)
x, delta, gamma, beta = rnd(R, C), rnd(R, C), rnd(C), rnd(C)
y, xo = torch.empty_like(x), torch.empty_like(x)
mean, rstd = torch.empty(R, device=dev), torch.empty(R, device=dev)
dx = torch.empty_like(x)
dg, db, dxs = (torch.empty(C, dtype=bf, device=dev) for _ in range(3))
fc, dact = rnd(R, 4 * C), rnd(R, 4 * C)
act, dbias = torch.empty_like(fc), torch.empty(4 * C, dtype=bf, device=dev)
ld = (V + 63) // 64 * 64
logits = rnd(R, ld, scale=2.0)
tgt = torch.randint(0, V, (R,), device=dev)
loss, hit = torch.empty(R, device=dev), torch.empty(R, device=dev)
attn_ops.causal_attention_fwd(qkv, out, lse, B, T, H)

res = {
    "attn_fwd": timed(lambda: attn_ops.causal_attention_fwd(qkv, out, lse, B, T, H)),
    "attn_bwd": timed(lambda: attn_ops.causal_attention_bwd(qkv, out, lse, dout, dqkv, scratch, B, T, H)),
    "ln_fwd_residual": timed(lambda: nn_ops.ln_fwd(x, gamma, beta, y, mean, rstd, delta=delta, x_out=xo)),
    "ln_bwd": timed(lambda: nn_ops.ln_bwd(dout, xo, gamma, mean, rstd, dx, dg, db, dres=delta, dxsum=dxs)),
    "gelu_fwd": timed(lambda: nn_ops.gelu_fwd(fc, act)),
    "gelu_bwd_colsum": timed(lambda: nn_ops.gelu_bwd_colsum(fc, dact, dact, dbias)),
    "colsum_2304": timed(lambda: nn_ops.colsum(qkv, torch.empty(3 * C, dtype=bf, device=dev))),
    "cross_entropy_fwd_bwd": timed(lambda: nn_ops.cross_entropy_fwd_bwd(logits, tgt, V, 1.0 / R, loss, hit), 4),
}
q, k, v = (t.transpose(1, 2) for t in qkv.view(B, T, 3, H, 64).unbind(2))
res["torch_sdpa_fwd"] = timed(lambda: F.scaled_dot_product_attention(q, k, v, is_causal=True))
xf = x.float()
res["torch_layer_norm_fp32"] = timed(lambda: F.layer_norm(xf, (C,), gamma.float(), beta.float()))
res["torch_gelu"] = timed(lambda: F.gelu(fc, approximate="tanh"))
lg = logits[:, :V]
res["torch_cross_entropy_fp32_fwd"] = timed(lambda: F.cross_entropy(lg.float(), tgt), 4)
flops = 4.0 * B * H * T * T * 64 / 2
res["attn_fwd_tflops"] = round(flops / res["attn_fwd"] / 1e6, 1)
res["attn_bwd_tflops"] = round(3.5 * flops / res["attn_bwd"] / 1e6, 1)
print(json.dumps(res))
