"""A few launches of every transformer kernel at GPT-2-small shapes (for rocprofv3).
    rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d out -- python tools/profile_transformer_ops.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nanofed_amd.ops import attention as attn_ops  # noqa: E402
from nanofed_amd.ops import nn_ops  # noqa: E402

B, T, H = 4, 1024, 12
C, R, V = H * 64, 4 * 1024, 50257
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
bf = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, generator=g, device=dev) * scale).to(bf)


qkv, dout = rnd(R, 3 * C), rnd(R, C)
out, dqkv = torch.empty(R, C, dtype=bf, device=dev), torch.empty(R, 3 * C, dtype=bf, device=dev)
lse = torch.empty(B, H, T, device=dev)
scratch = attn_ops.backward_scratch(B, H, T, dev)
x, delta, gamma, beta = rnd(R, C), rnd(R, C), rnd(C), rnd(C)
y, xo = torch.empty_like(x), torch.empty_like(x)
mean, rstd = torch.empty(R, device=dev), torch.empty(R, device=dev)
dx, dg, db, dxs = torch.empty_like(x), torch.empty(C, dtype=bf, device=dev), torch.empty(C, dtype=bf, device=dev), torch.empty(C, dtype=bf, device=dev)
fc, dact = rnd(R, 4 * C), rnd(R, 4 * C)
act, dbias = torch.empty_like(fc), torch.empty(4 * C, dtype=bf, device=dev)
ld = (V + 63) // 64 * 64
logits = rnd(R, ld, scale=2.0)
tgt = torch.randint(0, V, (R,), device=dev, generator=g)
loss, hit = torch.empty(R, device=dev), torch.empty(R, device=dev)
for _ in range(3):
    attn_ops.causal_attention_fwd(qkv, out, lse, B, T, H)
    attn_ops.causal_attention_bwd(qkv, out, lse, dout, dqkv, scratch, B, T, H)
    nn_ops.ln_fwd(x, gamma, beta, y, mean, rstd, delta=delta, x_out=xo)
    nn_ops.ln_bwd(dout, xo, gamma, mean, rstd, dx, dg, db, dres=delta, dxsum=dxs)
    nn_ops.gelu_fwd(fc, act)
    nn_ops.gelu_bwd_colsum(fc, dact, dact, dbias)
    nn_ops.colsum(qkv, torch.empty(3 * C, dtype=bf, device=dev))
    nn_ops.cross_entropy_fwd_bwd(logits, tgt, V, 1.0 / R, loss, hit)
torch.cuda.synchronize()
print("done")
