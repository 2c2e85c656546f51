"""A few launches of the 3x3 convolution kernels at the four ResNet-18 stage shapes
(for rocprofv3).
    rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d out -- python tools/profile_conv_ops.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nanofed_amd.ops import conv_ops  # noqa: E402

CL = torch.channels_last
torch.cuda.set_device(0)
g = torch.Generator(device="cuda").manual_seed(0)
for b, h, w, c in [(64, 32, 32, 64), (64, 16, 16, 128), (64, 8, 8, 256), (64, 4, 4, 512)]:
    x = torch.randn(b, c, h, w, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=CL)
    wt = (torch.randn(c, c, 3, 3, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
    wt = wt.contiguous(memory_format=CL)
    dy = torch.randn(b, c, h, w, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=CL)
    for _ in range(3):
        conv_ops.conv3x3_forward(x, wt)
        conv_ops.conv3x3_weight_grad(x, dy)
    torch.cuda.synchronize()
print("done")
