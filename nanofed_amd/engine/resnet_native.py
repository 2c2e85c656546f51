"""ResNet-18 local training with every client of the GPU in each kernel launch.

``TorchGroupTrainer`` steps one client at a time: a batch-64 ResNet layer is a small
kernel, the step is ~250 launches per client, and even four concurrent client streams
leave the chip waiting on launch ramps.  Here the batch of a launch is the
concatenation of the clients' batches - ``G`` groups of ``B`` images - and every kernel
picks the group's own weights, statistics and gradient slots by its workgroup index
(``nanofed_amd.ops.group_ops``).  A step for eight clients is the same ~250 launches as
a step for one, each over eight times the data.

The trainer is explicit: no autograd, no module in the training path.

* parameters: the fp32 arena rows are the masters, ``self._shadow`` (inherited) their
  bf16 copy that the kernels read, ``self._gradb`` a flat bf16 gradient buffer with the
  arena's row layout; one launch applies SGD to all rows and refreshes the copy;
* forward per convolution unit: convolution (implicit-GEMM kernel for 3x3 / stride 1;
  patch gather + batched GEMM for the stem, the strided 3x3 and the 1x1 shortcuts),
  then BatchNorm (+ residual) (+ ReLU) with the statistics of each group;
* backward in reverse, where a tensor that fed two branches receives its gradient as
  the sum of two tensors inside the consuming kernel (no add kernels).  Where the gradient
  of a BatchNorm's output comes out of the 3x3 data-gradient kernel, that kernel's
  epilogue already adds the second branch, applies the ReLU mask and leaves the
  per-channel partial sums, so the BatchNorm backward is one final sum and one pass over
  the pixels instead of two passes; below a downsampling block the scatter of the 1x1
  shortcut's data gradient does the same (16 of the 20 BatchNorm layers together;
  ``NANOFED_RESNET_FUSE_BN_BWD=0`` keeps the separate passes);
* the head (global average pooling, classifier, cross-entropy, their gradients, the
  loss / accuracy accumulators) is two launches;
* the whole step is captured once per group range into a hipGraph.

Steps in which some client has no full batch left (unequal shards, ragged tails) run the
contiguous runs of clients that do have one as smaller groups, and the ragged batch
itself through the inherited per-client path - same parameters, same bf16 copy.

The reference has no ResNet and trains one client per process
(nanofed/trainer/base.py:115-198); this trainer serves the convolution-heavy
configuration of the benchmark suite.
"""

from __future__ import annotations

import os
from dataclasses import dataclass

import torch
import torch.nn as nn

from nanofed_amd.core.exceptions import DeviceError
from nanofed_amd.models.resnet import BasicBlock, ResNet18
from nanofed_amd.ops import gemm_ops
from nanofed_amd.ops import group_ops as go
from nanofed_amd.ops.group_ops import Ref
from nanofed_amd.utils.streams import side_stream

from .config import EngineConfig
from .trainers import M_ACC, M_LOSS, M_SAMPLES, M_VALID, TorchGroupTrainer

BF16 = torch.bfloat16
STEM_K = 32
# clients per launch: larger groups are stepped in chunks (activation memory grows with
# the group, the kernels are full long before)
MAX_GROUP = int(os.environ.get("NANOFED_RESNET_GROUP", "16"))


@dataclass
class _Unit:
    """One convolution + BatchNorm of the network."""

    conv: str            # state-dict prefix of the convolution
    bn: str              # ... of its BatchNorm
    cin: int
    cout: int
    hin: int             # input height = width
    hout: int
    k: int
    stride: int
    pad: int
    kp: int = 0           # patch row length of the patch-gather + GEMM path


@dataclass
class _Block:
    u1: _Unit
    u2: _Unit
    us: _Unit | None


def supports(cfg: EngineConfig, model: nn.Module, device: torch.device, data=None) -> bool:
    """The grouped kernels cover the CIFAR ResNet-18 of this package (width 64) in bf16
    without DP, batch sizes that are multiples of 8, on any image set of the CIFAR shape:
    uint8 ``[N, 3, 32, 32]`` with one mean and one positive std per channel (``data``: the
    engine's ``FederatedData``; what is tested is the data, not the dataset's name)."""
    if os.environ.get("NANOFED_RESNET_NATIVE", "1") == "0":
        return False
    if device.type != "cuda" or cfg.dtype != "bf16" or cfg.dp is not None:
        return False
    if not isinstance(model, ResNet18) or model.fc.in_features != 512 or model.fc.out_features > 16:
        return False
    if not gemm_ops.enabled(device) or not cifar_shaped(data):
        return False
    return cfg.batch_size % 8 == 0 and 8 <= cfg.batch_size <= 512


def cifar_shaped(data) -> bool:
    if data is None or getattr(data, "kind", None) != "image":
        return False
    x = data.inputs
    if x.dtype != torch.uint8 or x.dim() != 4 or tuple(x.shape[1:]) != (3, 32, 32):
        return False
    if data.labels is None or len(data.mean) != 3 or len(data.std) != 3:
        return False
    return all(float(s) > 0.0 for s in data.std)


class ResNetNativeTrainer(TorchGroupTrainer):
    def __init__(self, cfg, pool, shards, model_factory, eval_inputs=None, eval_labels=None):
        super().__init__(cfg, pool, shards, model_factory, eval_inputs, eval_labels)
        if not self._direct or self._shadow is None:
            raise DeviceError("the grouped ResNet trainer needs bf16 training on a HIP device")
        model = self.modules[0]
        if not isinstance(model, ResNet18):
            raise DeviceError("the grouped ResNet trainer is built for models.resnet.ResNet18")
        if shards.inputs.dim() != 4 or tuple(shards.inputs.shape[1:]) != (3, 32, 32):
            raise DeviceError("the grouped ResNet trainer needs 3x32x32 images")
        self.layout = pool.layout
        self.B = cfg.batch_size
        self.classes = model.fc.out_features
        self._streams = None  # the clients share every launch: no client streams
        G = len(pool)
        self._gradb = torch.zeros(G, self.layout.trainable_numel, dtype=BF16, device=self.device)
        self._arena = pool.data  # [G, numel] fp32
        self._stem, self._blocks = self._plan(model)
        mean = [float(v) for v in shards.mean.flatten().tolist()]
        std = [float(v) for v in shards.std.flatten().tolist()]
        self._scale = [1.0 / (255.0 * s) for s in std]
        self._shift = [-m / s for m, s in zip(mean, std)]
        self._bn_eps = float(model.bn1.eps)
        self._bn_momentum = float(model.bn1.momentum)
        for m in model.modules():
            if isinstance(m, nn.BatchNorm2d) and (m.eps != self._bn_eps or m.momentum != self._bn_momentum):
                raise DeviceError("BatchNorm layers with different eps / momentum")
        self._gbufs: dict[int, dict] = {}
        self._ggraphs: dict[tuple[int, int], tuple[torch.cuda.CUDAGraph, torch.Tensor]] = {}
        self._acc3 = torch.zeros(G, 3, dtype=torch.float32, device=self.device)
        self._acc = self._acc3  # the inherited per-client path accumulates into the same rows
        self._lr = float(cfg.learning_rate)
        self._fuse_bn_bwd = os.environ.get("NANOFED_RESNET_FUSE_BN_BWD", "1") != "0"
        # recompute the ReLU mask of a BatchNorm without residual from its input (else read it)
        self._remask = os.environ.get("NANOFED_RESNET_REMASK", "1") != "0"
        # below a downsampling block: join the branches in the shortcut's scatter
        self._join = os.environ.get("NANOFED_RESNET_JOIN", "1") != "0"

    # -- network plan ---------------------------------------------------------------------
    def _plan(self, model: ResNet18) -> tuple[_Unit, list[_Block]]:
        def unit(conv_name: str, bn_name: str, conv: nn.Conv2d, hin: int) -> _Unit:
            k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
            hout = (hin + 2 * p - k) // s + 1
            u = _Unit(conv_name, bn_name, conv.in_channels, conv.out_channels, hin, hout, k, s, p)
            u.kp = -(-(k * k * u.cin) // 8) * 8
            return u

        stem = unit("conv1", "bn1", model.conv1, 32)
        stem.kp = STEM_K
        blocks: list[_Block] = []
        h = stem.hout
        for li in range(1, 5):
            for bi, blk in enumerate(getattr(model, f"layer{li}")):
                if not isinstance(blk, BasicBlock):
                    raise DeviceError("unexpected block type")
                pre = f"layer{li}.{bi}"
                u1 = unit(f"{pre}.conv1", f"{pre}.bn1", blk.conv1, h)
                u2 = unit(f"{pre}.conv2", f"{pre}.bn2", blk.conv2, u1.hout)
                us = None
                if len(blk.shortcut) > 0:
                    us = unit(f"{pre}.shortcut.0", f"{pre}.shortcut.1", blk.shortcut[0], h)
                blocks.append(_Block(u1, u2, us))
                h = u2.hout
        return stem, blocks

    def _units(self) -> list[_Unit]:
        out = [self._stem]
        for b in self._blocks:
            out += [b.u1, b.u2] + ([b.us] if b.us is not None else [])
        return out

    def _off(self, name: str) -> int:
        return self.layout[name].offset

    # -- buffers of a group size ----------------------------------------------------------
    def _buffers(self, Gn: int) -> dict:
        if Gn in self._gbufs:
            return self._gbufs[Gn]
        dev, B = self.device, self.B
        N = Gn * B

        def bf(*shape):
            return torch.empty(shape, dtype=BF16, device=dev)

        def f32(*shape):
            return torch.empty(shape, dtype=torch.float32, device=dev)

        bufs: dict = {"units": {}, "blocks": []}
        ws_fwd = ws_bwd = ws_wgrad = ws_pre = slabs = 0
        for u in self._units():
            ho, C = u.hout, u.cout
            d = {
                "y": bf(N, ho, ho, C), "out": bf(N, ho, ho, C), "dyp": bf(N, ho, ho, C),
                "stats": f32(Gn, 2, C),
            }
            # implicit-GEMM 3x3 kernel where it covers the shape, else patch gather + GEMM
            d["native"] = (u.k == 3 and u.stride == 1 and u.pad == 1 and u.cin % 64 == 0
                           and go.conv3x3_supported(Gn, B, u.hin, u.hin, u.cin, u.cout))
            ws_bwd = max(ws_bwd, go.bn_workspace_floats(Gn, C))
            if d["native"]:
                tiles = go.conv3x3_tiles(Gn, B, u.hin, u.hin, C)
                d["tiles"] = tiles
                ws_fwd = max(ws_fwd, Gn * tiles * 2 * C)
                ws_wgrad = max(ws_wgrad, go.conv3x3_wgrad_workspace(Gn, B, u.hin, u.hin, u.cin, C))
                d["dx"] = bf(N, u.hin, u.hin, u.cin)
                # pixel tiles per group of the data-gradient launch (its output has cin channels)
                d["dtiles"] = go.conv3x3_tiles(Gn, B, u.hin, u.hin, u.cin)
                ws_pre = max(ws_pre, Gn * d["dtiles"] * 2 * u.cin)
            else:
                M = B * ho * ho
                d["col"] = bf(N * ho * ho, u.kp)
                ws_fwd = max(ws_fwd, go.bn_workspace_floats(Gn, C))
                slabs = max(slabs, go.gemm_slab_floats(M, C, u.kp, gemm_ops.NT, Gn),
                            go.gemm_slab_floats(C, u.kp, M, gemm_ops.TN, Gn))
                if u is not self._stem:
                    if u.cin % 8:
                        raise DeviceError("data gradients need channel counts that are multiples of 8")
                    d["dcol"] = bf(N * ho * ho, u.kp)
                    d["dx"] = bf(N, u.hin, u.hin, u.cin)
                    # rows of partial sums per group when the scatter joins the branches
                    d["jblocks"] = (go.col2im_join_blocks(B, u.hin, u.hin, u.cin)
                                    if 256 % (u.cin // 8) == 0 else 0)
                    ws_pre = max(ws_pre, Gn * d["jblocks"] * 2 * u.cin)
                    slabs = max(slabs, go.gemm_slab_floats(M, u.kp, C, gemm_ops.NN, Gn))
            bufs["units"][u.conv] = d
        for blk in self._blocks:
            ho, C = blk.u2.hout, blk.u2.cout
            bufs["blocks"].append({"dres": bf(N, ho, ho, C)})
        bufs["ws_fwd"] = f32(max(ws_fwd, 1))
        bufs["ws_bwd"] = f32(max(ws_bwd, 1))
        bufs["ws_pre"] = f32(max(ws_pre, 1))
        bufs["ws_wgrad"] = bf(max(ws_wgrad, 8))
        bufs["slabs"] = f32(max(slabs, 1))
        bufs["wpad"] = torch.zeros(Gn, 64, STEM_K, dtype=BF16, device=dev)
        bufs["dwpad"] = bf(Gn, 64, STEM_K)
        last = self._blocks[-1].u2
        bufs["pooled"] = bf(N, 512)
        bufs["dlogits"] = f32(N, 16)
        bufs["loss_rows"] = f32(N)
        bufs["correct_rows"] = f32(N)
        bufs["dhead"] = bf(N, last.hout, last.hout, last.cout)
        self._gbufs[Gn] = bufs
        return bufs

    # -- references into the per-client rows ------------------------------------------------
    def _w(self, g0: int, name: str) -> Ref:       # bf16 copy of a parameter
        return Ref(self._shadow[g0:], self._off(name))

    def _p32(self, g0: int, name: str) -> Ref:     # fp32 master (BatchNorm vectors, biases)
        return Ref(self._arena[g0:], self._off(name))

    def _g(self, g0: int, name: str) -> Ref:       # gradient slot
        return Ref(self._gradb[g0:], self._off(name))

    # -- one convolution unit ---------------------------------------------------------------
    def _unit_fwd(self, u: _Unit, x: torch.Tensor | None, res: torch.Tensor | None, relu: bool,
                  g0: int, Gn: int, bufs: dict) -> torch.Tensor:
        d = bufs["units"][u.conv]
        B, C = self.B, u.cout
        N = Gn * B
        rows = B * u.hout * u.hout
        w = self._w(g0, u.conv + ".weight")
        pre = 0
        if d["native"]:
            go.conv3x3_fwd(x, w, d["y"], bufs["ws_fwd"], Gn, B, u.hin, u.hin, u.cin, C)
            pre = d["tiles"]
        else:
            if u is self._stem:
                b_ptr, ldb, sb = bufs["wpad"], STEM_K, 64 * STEM_K
            else:
                go.im2col(x, d["col"], N, u.hin, u.hin, u.cin, u.hout, u.hout, u.k, u.stride,
                          u.pad, u.kp)
                b_ptr, ldb, sb = w, u.kp, w.stride
            go.gemm_batched(gemm_ops.NT, d["col"], b_ptr, d["y"], rows, C,
                            u.kp, u.kp, ldb, C, Gn, rows * u.kp, sb, rows * C, bufs["slabs"],
                            go._stream(d["y"]))
        go.bn_fwd(d["y"], res, d["out"], self._p32(g0, u.bn + ".weight"),
                  self._p32(g0, u.bn + ".bias"), self._p32(g0, u.bn + ".running_mean"),
                  self._p32(g0, u.bn + ".running_var"), d["stats"], bufs["ws_fwd"], Gn, rows, C,
                  relu, self._bn_momentum, self._bn_eps, pre)
        return d["out"]

    def _unit_bwd(self, u: _Unit, x: torch.Tensor | None, dy: torch.Tensor,
                  dy2: torch.Tensor | None, dres: torch.Tensor | None, relu: bool, g0: int,
                  Gn: int, bufs: dict, pre: int = 0, consumer: _Unit | None = None,
                  add: torch.Tensor | None = None, plain: bool = False) -> torch.Tensor | None:
        """Backward of BatchNorm and convolution; returns d(input) (None for the stem).

        ``pre`` > 0: ``dy`` is the final gradient of the BatchNorm's output (masked, joined)
        and ``bufs["ws_pre"]`` holds ``pre`` rows of partial sums per group, both left by
        the data-gradient launch that produced it.  ``consumer``: the unit whose BatchNorm
        output is this convolution's input; the data gradient then does that BatchNorm's
        masking and partial sums (with ``add``, the other branch's gradient) and what is
        returned is ``consumer``'s final gradient; ``plain``: that BatchNorm had no
        residual input (its ReLU mask is recomputed from its input instead of read)."""
        d = bufs["units"][u.conv]
        B, C = self.B, u.cout
        N = Gn * B
        rows = B * u.hout * u.hout
        if pre:
            go.bn_bwd_pre(dy, d["y"], d["stats"], self._p32(g0, u.bn + ".weight"), d["dyp"],
                          self._g(g0, u.bn + ".weight"), self._g(g0, u.bn + ".bias"),
                          bufs["ws_pre"], pre, bufs["ws_bwd"], Gn, rows, C)
        else:
            go.bn_bwd(dy, dy2, d["out"], d["y"], d["stats"], self._p32(g0, u.bn + ".weight"),
                      d["dyp"], dres, self._g(g0, u.bn + ".weight"), self._g(g0, u.bn + ".bias"),
                      bufs["ws_bwd"], Gn, rows, C, relu)
        w = self._w(g0, u.conv + ".weight")
        dw = self._g(g0, u.conv + ".weight")
        if d["native"]:
            go.conv3x3_wgrad(x, d["dyp"], dw, bufs["ws_wgrad"], Gn, B, u.hin, u.hin, u.cin, C)
            fuse = None
            if consumer is not None:
                cd = bufs["units"][consumer.conv]
                if plain and self._remask:
                    fuse = go.BnBwdFuse(None, cd["y"], add, bufs["ws_pre"], cd["stats"],
                                        self._p32(g0, consumer.bn + ".weight"),
                                        self._p32(g0, consumer.bn + ".bias"))
                else:
                    fuse = go.BnBwdFuse(cd["out"], cd["y"], add, bufs["ws_pre"])
            go.conv3x3_dgrad(d["dyp"], w, d["dx"], Gn, B, u.hin, u.hin, u.cin, C, fuse=fuse)
            return d["dx"]
        stream = go._stream(d["dyp"])
        if u is self._stem:
            c_ptr, ldc, sc = bufs["dwpad"], STEM_K, 64 * STEM_K
        else:
            c_ptr, ldc, sc = dw, u.kp, dw.stride
        go.gemm_batched(gemm_ops.TN, d["dyp"], d["col"], c_ptr, C, u.kp, rows,
                        C, u.kp, ldc, Gn, rows * C, rows * u.kp, sc, bufs["slabs"], stream)
        if u is self._stem:
            k = 9 * u.cin
            off = self._off(u.conv + ".weight")
            self._gradb[g0 : g0 + Gn, off : off + 64 * k].view(Gn, 64, k).copy_(
                bufs["dwpad"][:, :, :k])
            return None
        go.gemm_batched(gemm_ops.NN, d["dyp"], w, d["dcol"], rows,
                        u.kp, C, C, u.kp, u.kp, Gn, rows * C, w.stride, rows * u.kp, bufs["slabs"],
                        stream)
        if consumer is not None:
            cd = bufs["units"][consumer.conv]
            go.col2im_join(d["dcol"], d["dx"], add, cd["out"], cd["y"], bufs["ws_pre"], Gn, B,
                           u.hin, u.hin, u.cin, u.hout, u.hout, u.k, u.stride, u.pad, u.kp)
        else:
            go.col2im(d["dcol"], d["dx"], N, u.hin, u.hin, u.cin, u.hout, u.hout, u.k, u.stride,
                      u.pad, u.kp)
        return d["dx"]

    # -- one optimisation step of clients g0 .. g0 + Gn - 1 ----------------------------------
    def _forward(self, g0: int, Gn: int, rows: torch.Tensor, bufs: dict) -> torch.Tensor:
        stem = self._stem
        k = 9 * stem.cin
        off = self._off("conv1.weight")
        bufs["wpad"][:, :, :k].copy_(self._shadow[g0 : g0 + Gn, off : off + 64 * k].view(Gn, 64, k))
        go.stem_patches(self.shards.inputs, rows, bufs["units"]["conv1"]["col"], 32, 32,
                        self._scale, self._shift)
        x = self._unit_fwd(stem, None, None, True, g0, Gn, bufs)
        for blk in self._blocks:
            o1 = self._unit_fwd(blk.u1, x, None, True, g0, Gn, bufs)
            short = x if blk.us is None else self._unit_fwd(blk.us, x, None, False, g0, Gn, bufs)
            x = self._unit_fwd(blk.u2, o1, short, True, g0, Gn, bufs)
        return x

    def _native_step(self, g0: int, Gn: int, rows: torch.Tensor) -> None:
        cfg, B = self.cfg, self.B
        bufs = self._buffers(Gn)
        labels = self.shards.labels[rows]
        feat = self._forward(g0, Gn, rows, bufs)
        last = self._blocks[-1].u2
        P = last.hout * last.hout
        go.pool_fc_ce(feat, self._w(g0, "fc.weight"), self._p32(g0, "fc.bias"), labels,
                      bufs["pooled"], bufs["dlogits"], bufs["loss_rows"], bufs["correct_rows"],
                      bufs["dhead"], Gn, B, P, 512, self.classes)
        go.fc_grads(bufs["pooled"], bufs["dlogits"], bufs["loss_rows"], bufs["correct_rows"],
                    self._g(g0, "fc.weight"), self._g(g0, "fc.bias"), self._acc3[g0:], Gn, B, 512,
                    self.classes)
        # da (+ db): gradient of the block's output; pre > 0: da is final (see _unit_bwd)
        da, db, pre = bufs["dhead"], None, 0
        fuse = self._fuse_bn_bwd
        units = bufs["units"]
        for bi in reversed(range(len(self._blocks))):
            blk = self._blocks[bi]
            below = self._blocks[bi - 1].u2 if bi > 0 else self._stem  # produced the block's input
            x_in = units[below.conv]["out"]
            o1 = units[blk.u1.conv]["out"]
            dres = None if pre else bufs["blocks"][bi]["dres"]
            inner = fuse and units[blk.u2.conv]["native"]
            d_o1 = self._unit_bwd(blk.u2, o1, da, db, dres, True, g0, Gn, bufs, pre=pre,
                                  consumer=blk.u1 if inner else None, plain=True)
            gout = da if pre else dres  # masked, joined gradient of the block's output
            pre1 = units[blk.u2.conv]["dtiles"] if inner else 0
            # the data gradient of u1 ends at the block's input, together with the shortcut's
            joins = fuse and blk.us is None and units[blk.u1.conv]["native"]
            dxa = self._unit_bwd(blk.u1, x_in, d_o1, None, None, True, g0, Gn, bufs, pre=pre1,
                                 consumer=below if joins else None, add=gout if joins else None,
                                 plain=below is self._stem)
            if joins:
                da, db, pre = dxa, None, units[blk.u1.conv]["dtiles"]
            elif blk.us is not None:
                # the shortcut's scatter adds the main branch, masks and leaves the sums
                jb = units[blk.us.conv]["jblocks"] if fuse and self._join else 0
                dxb = self._unit_bwd(blk.us, x_in, gout, None, None, False, g0, Gn, bufs,
                                     consumer=below if jb else None, add=dxa if jb else None)
                da, db, pre = (dxb, None, jb) if jb else (dxa, dxb, 0)
            else:
                da, db, pre = dxa, gout, 0
        self._unit_bwd(self._stem, None, da, db, None, True, g0, Gn, bufs, pre=pre)
        go.sgd_rows(self._arena[g0 : g0 + Gn], self._gradb[g0 : g0 + Gn],
                    self._shadow[g0 : g0 + Gn], self.layout.trainable_numel, self._lr,
                    cfg.weight_decay)

    def _group_graph(self, g0: int, Gn: int) -> tuple[torch.cuda.CUDAGraph, torch.Tensor]:
        key = (g0, Gn)
        if key in self._ggraphs:
            return self._ggraphs[key]
        N = Gn * self.B
        static_rows = torch.cat([
            torch.arange(self.B, device=self.device) + int(self.shards.offsets[g0 + j])
            for j in range(Gn)
        ])
        assert static_rows.numel() == N
        keep = self._arena[g0 : g0 + Gn].clone()
        keep_acc = self._acc3[g0 : g0 + Gn].clone()
        side = side_stream(self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._native_step(g0, Gn, static_rows)  # first-call setup of the kernels
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self._native_step(g0, Gn, static_rows)
        self._arena[g0 : g0 + Gn].copy_(keep)
        self._acc3[g0 : g0 + Gn].copy_(keep_acc)
        self._shadow[g0 : g0 + Gn].copy_(self._arena[g0 : g0 + Gn, : self.layout.trainable_numel])
        self._ggraphs[key] = (graph, static_rows)
        return self._ggraphs[key]

    def _group_step(self, g0: int, Gn: int, rows: torch.Tensor) -> None:
        if self._use_graph:
            graph, static_rows = self._group_graph(g0, Gn)
            static_rows.copy_(rows)
            graph.replay()
        else:
            self._native_step(g0, Gn, rows)

    # -- a round ----------------------------------------------------------------------------
    def _train_round(self, round_idx: int) -> torch.Tensor:
        cfg, B = self.cfg, self.B
        G = len(self.pool)
        self.metrics.zero_()
        self._refresh_shadow()
        steps = [self.shards.steps_per_epoch(k) for k in range(G)]
        full = [self.shards.sizes[k] // B for k in range(G)]
        for epoch in range(cfg.local_epochs):
            self._acc3.zero_()
            perms = [self.shards.epoch_permutation(k, round_idx, epoch) for k in range(G)]
            stacked = None
            if len(set(self.shards.sizes)) == 1 and full[0] > 0:
                stacked = torch.stack([p[: full[0] * B] for p in perms]).view(G, full[0], B)
            for s in range(max(steps, default=0)):
                if stacked is not None and s < full[0]:
                    for g0 in range(0, G, MAX_GROUP):
                        Gn = min(MAX_GROUP, G - g0)
                        self._group_step(g0, Gn, stacked[g0 : g0 + Gn, s].reshape(-1))
                else:
                    for g0, Gn in _runs([k for k in range(G) if s < full[k]], MAX_GROUP):
                        rows = torch.cat([perms[k][s * B : (s + 1) * B] for k in range(g0, g0 + Gn)])
                        self._group_step(g0, Gn, rows)
                for k in range(G):
                    if full[k] <= s < steps[k]:  # the ragged last batch of client k
                        self._eager_step(k, perms[k][s * B :])
            # BatchNorm step counters (kept in the arena rows, see TorchGroupTrainer)
            if self._count_idx.numel():
                add = torch.tensor([float(n) for n in steps], device=self.device)
                self._arena[:, self._count_idx] += add[:, None]
            del perms, stacked
        batches = self._acc3[:, 2].clamp(min=1.0)
        self.metrics[:, M_SAMPLES] = torch.tensor(
            [float(n) for n in self.shards.sizes], dtype=torch.float32).to(self.device)
        self.metrics[:, M_LOSS] = self._acc3[:, 0] / batches
        self.metrics[:, M_ACC] = self._acc3[:, 1] / batches
        self.metrics[:, M_VALID] = 1.0
        return self.metrics

    # -- gradients of one batch per client, without an update (kernel tests) -----------------
    def loss_and_grads(self, rows: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        """Runs forward and backward for ``rows`` (``G * B`` row indices, client-major) and
        returns (per-client mean loss, the bf16 gradient buffer ``[G, trainable]``).  The
        parameters and BatchNorm running statistics are put back afterwards."""
        G = len(self.pool)
        keep = self._arena.clone()
        keep_acc = self._acc3.clone()
        self._refresh_shadow()
        self._acc3.zero_()
        lr, self._lr = self._lr, 0.0
        try:
            self._native_step(0, G, rows)
        finally:
            self._lr = lr
        loss = self._acc3[:, 0].clone()
        self._arena.copy_(keep)
        self._acc3.copy_(keep_acc)
        self._refresh_shadow()
        return loss, self._gradb


def _runs(ks: list[int], longest: int = 1 << 30) -> list[tuple[int, int]]:
    """Runs of consecutive integers as (first, length), none longer than ``longest``."""
    out: list[tuple[int, int]] = []
    for k in ks:
        if out and out[-1][0] + out[-1][1] == k and out[-1][1] < longest:
            out[-1] = (out[-1][0], out[-1][1] + 1)
        else:
            out.append((k, 1))
    return out
