"""Microseconds for the fused BatchNorm(+residual)+ReLU forward+backward at the
ResNet-18 shapes (batch 64, 32x32 inputs) next to PyTorch's batch_norm + add + relu."""
import json
import os
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nanofed_amd.ops.bn_ops import batch_norm_act  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) * 1000.0 / reps, 1)


out = {}
for shape in [(64, 64, 32, 32), (64, 128, 16, 16), (64, 256, 8, 8), (64, 512, 4, 4)]:
    c = shape[1]
    bn = nn.BatchNorm2d(c).to(dev)
    x = torch.randn(shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    r = torch.randn(shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    dy = torch.randn(shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)

    def fused():
        batch_norm_act(bn, x, residual=r).backward(dy)

    def eager():
        F.relu(bn(x) + r).backward(dy)

    out[str(shape)] = {"fused_us": timed(fused), "torch_us": timed(eager)}
print(json.dumps(out))
