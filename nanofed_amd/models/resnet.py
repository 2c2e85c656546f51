"""ResNet-18 for 32x32 inputs (CIFAR-10 variant: 3x3 stem, no stem pooling).

Used for the conv-heavy federated configuration.  BatchNorm running
statistics and ``num_batches_tracked`` are buffers; the arena layout keeps
integer buffers out of the averaged float region (see
``nanofed_amd.runtime.arena.ArenaLayout``).
"""

from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from nanofed_amd.ops.bn_ops import batch_norm_act
from nanofed_amd.ops.conv_ops import conv2d


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, in_planes: int, planes: int, stride: int = 1) -> None:
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.shortcut: nn.Module = nn.Sequential()
        if stride != 1 or in_planes != planes:
            self.shortcut = nn.Sequential(
                nn.Conv2d(in_planes, planes, 1, stride, bias=False),
                nn.BatchNorm2d(planes),
            )

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        # BatchNorm, the residual add and the ReLU are one fused op on the GPU; the
        # stride-1 3x3 convolutions run on the implicit-GEMM MFMA kernels there, the
        # strided, 1x1 and stem convolutions as patch gather + GEMM
        y, sums = conv2d(self.conv1, x, with_sums=True)  # sums: BatchNorm's statistics
        out = batch_norm_act(self.bn1, y, partials=sums)
        if len(self.shortcut) == 0:
            short = x
        else:
            short = batch_norm_act(self.shortcut[1], conv2d(self.shortcut[0], x), relu=False)
        y, sums = conv2d(self.conv2, out, with_sums=True)
        return batch_norm_act(self.bn2, y, residual=short, partials=sums)


class ResNet18(nn.Module):
    def __init__(self, num_classes: int = 10, width: int = 64) -> None:
        super().__init__()
        self.in_planes = width
        self.conv1 = nn.Conv2d(3, width, 3, 1, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.layer1 = self._stage(width, 2, 1)
        self.layer2 = self._stage(width * 2, 2, 2)
        self.layer3 = self._stage(width * 4, 2, 2)
        self.layer4 = self._stage(width * 8, 2, 2)
        self.fc = nn.Linear(width * 8, num_classes)

    def _stage(self, planes: int, blocks: int, stride: int) -> nn.Sequential:
        layers = []
        for s in [stride] + [1] * (blocks - 1):
            layers.append(BasicBlock(self.in_planes, planes, s))
            self.in_planes = planes
        return nn.Sequential(*layers)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = batch_norm_act(self.bn1, conv2d(self.conv1, x))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = torch.flatten(F.adaptive_avg_pool2d(x, 1), 1)
        return self.fc(x)
