"""Native (hand-written HIP) local trainer for the MNIST CNN.

Implements the ``GroupTrainer`` interface on top of csrc/hip/mnist_cnn.hip:
the simulated clients of a GPU advance one SGD step per chain of eight or nine
kernel launches, each kernel working on several clients at once (blockIdx.y),
and one local epoch is recorded once in hipGraphs and replayed - the index
table, the valid-sample table and the dropout step counter live in device
memory, so a replay trains on the next epoch's shuffled batches without any
host involvement.  With four or more clients on the GPU the clients are split
into pairs whose chains run on concurrent streams (views of one set of buffers).

Training semantics are the reference's (nanofed/trainer/base.py:134-198 with
nanofed/models/mnist.py:16-28 and nanofed/trainer/torch.py:14): per batch
forward, mean cross-entropy, backward, plain SGD; the partial last batch of a
shard is kept (``drop_last=False``) by giving its padding rows zero loss
weight.  Metrics are the mean loss / accuracy over the batches of the last
local epoch and the shard size.

Parameter storage differs from PyTorch's for two tensors (conv2.weight as
[oc][kh][kw][c], fc1.weight with NHWC-flattened columns); ``make_layout``
describes that with arena storage permutations, so ``state_dict`` /
``load_state_dict`` of the arenas stay in PyTorch order.
"""

from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_long, c_ulonglong, c_void_p

import numpy as np
import torch
import torch.nn as nn

from nanofed_amd.core.exceptions import DeviceError
from nanofed_amd.ops import library
from nanofed_amd.runtime.arena import ArenaLayout, ArenaPool

from .config import EngineConfig
from .data import DeviceShards, FederatedData
from .trainers import (
    M_ACC,
    M_EPSILON,
    M_LOSS,
    M_SAMPLES,
    M_VALID,
    METRIC_COLUMNS,
    GroupTrainer,
)

BATCH = 64
FC1_SPLITS = 32
CONV2_WG_PER_CLIENT = 32

STAGES = {
    "conv1_fwd": 0, "conv2_fwd": 1, "fc1_fwd": 2, "head": 3, "fc1_dgrad": 4,
    "fc1_wgrad": 5, "conv2_dgrad": 6, "conv2_wgrad": 7, "finalize": 8,
    "sgd_pack": 9, "pack": 10, "dp_scale": 11, "finalize_sgd": 12, "bwd_pair": 13,
    "bwd_triple": 16, "conv12_fwd": 17, "dp_pair": 18, "dp_clip": 19,
}


class MnistCtx(ctypes.Structure):
    """Mirror of ``struct MnistCtx`` in csrc/hip/mnist_cnn.hip."""

    _fields_ = [
        ("G", c_int), ("B", c_int), ("S", c_int), ("WPC", c_int),
        ("train", c_int), ("client_base", c_int), ("fuse_sgd", c_int), ("dp_pass", c_int),
        ("split", c_int), ("conv1_split", c_int),
        ("images", c_void_p), ("labels", c_void_p),
        ("idx", c_void_p), ("idx_stride", c_long),
        ("nvalid", c_void_p), ("nvalid_stride", c_long),
        ("arena", c_void_p), ("arena_stride", c_long),
        ("grad", c_void_p), ("grad_stride", c_long),
        ("shadow", c_void_p), ("shadow_stride", c_long),
        ("w2t", c_void_p), ("w2t_stride", c_long),
        ("n_arena", c_long),
        ("off_c1w", c_int), ("off_c1b", c_int), ("off_c2w", c_int), ("off_c2b", c_int),
        ("off_f1w", c_int), ("off_f1b", c_int), ("off_f2w", c_int), ("off_f2b", c_int),
        ("a1", c_void_p), ("p2", c_void_p), ("pmask", c_void_p),
        ("fc1_slabs", c_void_p), ("dh1", c_void_p), ("dp2", c_void_p), ("a1mask", c_void_p),
        ("conv1_part", c_void_p), ("conv2_part", c_void_p), ("c2b_part", c_void_p),
        ("head_part", c_void_p),
        ("metrics", c_void_p), ("eval_out", c_void_p),
        ("step_counter", c_void_p), ("seed", c_ulonglong),
        ("lr", c_float), ("wd", c_float), ("mean", c_float), ("inv_std", c_float),
        ("p_drop1", c_float), ("p_drop2", c_float),
        ("dp_clip", c_float), ("dp_noise_std", c_float),
        ("dp_normsq", c_void_p), ("dp_rowscale", c_void_p), ("dp_c2b", c_void_p),
        ("ragged", c_int), ("cur_step", c_int),
        ("micro", c_int), ("accumulate", c_int), ("ntotal", c_void_p),
        ("bwd_triple", c_int),
        ("images_f32", c_void_p),
    ]


def _lib():
    lib = library.hip_lib()
    if not getattr(lib, "_mnist_declared", False):
        P = ctypes.POINTER(MnistCtx)
        lib.nf_mnist_ctx_size.restype = c_int
        lib.nf_mnist_stage.argtypes = [P, c_int, c_int, c_int, c_void_p]
        lib.nf_mnist_train_step.argtypes = [P, c_int, c_int, c_void_p]
        lib.nf_mnist_train_steps.argtypes = [P, c_int, c_int, c_void_p]
        lib.nf_mnist_eval_step.argtypes = [P, c_int, c_int, c_void_p]
        lib.nf_mnist_pack.argtypes = [P, c_void_p]
        for fn in ("nf_mnist_stage", "nf_mnist_train_step", "nf_mnist_train_steps",
                   "nf_mnist_eval_step", "nf_mnist_pack"):
            getattr(lib, fn).restype = c_int
        if lib.nf_mnist_ctx_size() != ctypes.sizeof(MnistCtx):
            raise DeviceError(
                "MnistCtx layout mismatch between Python and the HIP library: "
                f"{ctypes.sizeof(MnistCtx)} vs {lib.nf_mnist_ctx_size()}"
            )
        lib._mnist_declared = True
    return lib


MAX_MICRO = 16  # batch sizes up to 16 * 64


def micro_batches(batch_size: int) -> int:
    """Micro-batches of up to ``BATCH`` rows that make up one batch."""
    return max(1, -(-batch_size // BATCH))


def supports(cfg: EngineConfig) -> bool:
    """True when the hand-written kernels cover ``cfg`` (the engine's ``auto``
    backend then takes this trainer instead of the PyTorch one).  The kernels
    work on 64 rows at a time: a smaller batch is one partly filled pass, a
    larger one several passes whose gradients are added up before the SGD
    update.  DP-SGD is built for the batch of 64."""
    if cfg.model != "mnist_cnn" or set(cfg.model_kwargs) - {"dropout"}:
        return False
    if cfg.dp is not None:
        return cfg.dp.mode == "per_sample" and cfg.batch_size == BATCH
    return 1 <= cfg.batch_size <= MAX_MICRO * BATCH


def native_permutations() -> dict[str, torch.Tensor]:
    """stored_flat = canonical_flat[perm] for the two re-ordered tensors."""
    conv2 = torch.arange(64 * 32 * 9).view(64, 32, 3, 3).permute(0, 2, 3, 1).reshape(-1)
    fc1 = torch.arange(128 * 9216).view(128, 64, 144).permute(0, 2, 1).reshape(-1)
    return {"conv2.weight": conv2, "fc1.weight": fc1}


def make_layout(module: nn.Module) -> ArenaLayout:
    return ArenaLayout.from_module(module).with_perms(native_permutations())


class NativeBuffers:
    """Activations, gradients-in-flight and partial slabs for G concurrent
    client batches, plus the ``MnistCtx`` that points at them."""

    def __init__(
        self,
        layout: ArenaLayout,
        G: int,
        device: torch.device,
        images: torch.Tensor,
        labels: torch.Tensor,
        arena: torch.Tensor,
        arena_stride: int,
        grad: torch.Tensor | None,
        max_steps: int,
        mean: float,
        std: float,
        cfg: EngineConfig | None = None,
        client_base: int = 0,
        train: bool = True,
        seed: int = 0,
        fuse_sgd: bool = True,
        dp=None,
        parent: "NativeBuffers | None" = None,
        first: int = 0,
        micro: int = 1,
    ) -> None:
        """``parent`` / ``first``: build a view of clients [first, first + G) of an
        existing set of buffers - same memory (batches, activations, counters,
        metrics), own kernel geometry and own conv2 partial slabs - so that groups
        of clients can run as separate launches."""
        if device.type != "cuda":
            raise DeviceError("the native MNIST trainer needs a HIP device")
        B = BATCH
        self.G, self.device, self.max_steps = G, device, max(1, max_steps)
        self.micro = micro
        if micro > 1 and (dp is not None or not train):
            raise DeviceError("micro-batch accumulation is for plain SGD training")
        table_steps = self.max_steps * micro

        def z(*shape, dtype=torch.float32, name=None):
            if parent is not None and name is not None:
                return getattr(parent, name)[first : first + G]
            return torch.zeros(*shape, dtype=dtype, device=device)

        self.images = images.reshape(images.shape[0], -1).contiguous()
        if self.images.dtype != torch.uint8 or self.images.shape[1] != 784:
            raise DeviceError("native MNIST trainer expects uint8 [N, 784] images")
        self.labels = labels.to(device=device, dtype=torch.int32).contiguous()
        self.idx = z(G, table_steps * B, dtype=torch.int32, name="idx")
        self.nvalid = z(G, table_steps, dtype=torch.int32, name="nvalid")
        self.ntotal = z(G, table_steps, dtype=torch.int32, name="ntotal")
        self.shadow = z(G, layout.numel, dtype=torch.bfloat16, name="shadow")
        self.w2t = z(G, 9 * 32 * 64, dtype=torch.bfloat16, name="w2t")
        self.a1 = z(G, B, 676, 32, dtype=torch.bfloat16, name="a1")
        self.p2 = z(G, B, 144, 64, dtype=torch.bfloat16, name="p2")
        self.pmask = z(G, B, 144, 64, dtype=torch.uint8, name="pmask")
        self.fc1_slabs = z(G, FC1_SPLITS, B, 128, name="fc1_slabs")
        self.dh1 = z(G, B, 128, dtype=torch.bfloat16, name="dh1")
        self.dp2 = z(G, B, 144, 64, dtype=torch.bfloat16, name="dp2")
        self.a1mask = z(G, B, 676, dtype=torch.int32, name="a1mask")
        self.conv1_part = z(G, 3 * B, 320, name="conv1_part")
        # Kernel geometry by the number of clients on the GPU (measured with
        # tools/bench_mnist_stages.py): with few clients the conv2 kernels are cut into
        # one band per workgroup (3x the workgroups, 1/3 the work each), conv1 into 2-4
        # pieces per image and conv2_wgrad into more, smaller workgroups, so the chip
        # is not left idle.  DP-SGD measures per-sample norms from per-image tiles:
        # one image per workgroup there.
        # A group of clients that shares the GPU with other groups (parent given) is
        # sized by the number of clients on the GPU, not by its own: what matters is
        # how full the chip is.
        total = parent.G if parent is not None else G
        if dp is not None:
            split, wpc, conv1_split = 1, B, 1
        elif total <= 1:
            split, wpc, conv1_split = 3, 3 * B, 4
        elif total == 2:
            split, wpc, conv1_split = 3, 96, 4
        elif total <= 4:
            split, wpc, conv1_split = 3, 64, 2
        elif parent is not None:
            # pairs of clients on concurrent streams (measured for 8 clients: 36.5
            # rounds/s against 34.1 with the two-client row above; weight-gradient
            # workgroups per client, with the three backward kernels in one launch: 32 ->
            # 45.2 rounds/s, 48 -> 45.0, 64 -> 44.7, 96 -> 42.0 -
            # tools/probes/geometry_resweep.sh)
            split, wpc, conv1_split = 1, 32, 1
        else:
            split, wpc, conv1_split = 1, CONV2_WG_PER_CLIENT, 1
        if cfg is not None and cfg.conv_split in (1, 3) and dp is None:
            split = cfg.conv_split
        # one geometry for every client count: the partial sums of the convolution
        # gradients are then associated the same way however the clients are placed
        invariant = cfg is not None and cfg.placement_invariant and dp is None
        if invariant:
            split, wpc, conv1_split = 1, 64, 1
        # tuning override: NANOFED_MNIST_GEOMETRY="split,wpc,conv1_split"
        geometry = os.environ.get("NANOFED_MNIST_GEOMETRY")
        if geometry and dp is None and not (cfg is not None and cfg.placement_invariant):
            split, wpc, conv1_split = (int(v) for v in geometry.split(","))
        self.conv2_part = z(G, wpc, 18432)
        self.dp_normsq = z(G, B, 4, name="dp_normsq")
        self.dp_rowscale = z(G, B, name="dp_rowscale")
        self.dp_c2b = z(G, B, 64, name="dp_c2b")
        self.c2b_part = z(G, 144, 64, name="c2b_part")
        self.head_part = z(G, B // 8, 1440, name="head_part")
        self.metrics = z(G, 4, name="metrics")
        self.eval_out = z(G, B // 8, 2)
        self.step_counter = z(G, dtype=torch.int64, name="step_counter")
        self.grad = grad if grad is not None else z(1, layout.trainable_numel)
        self.arena = arena

        c = MnistCtx()
        c.G, c.B, c.S, c.WPC = G, B, FC1_SPLITS, wpc
        c.split = split
        c.conv1_split = conv1_split
        c.train = 1 if train else 0
        c.client_base = client_base
        c.images = self.images.data_ptr()
        c.labels = self.labels.data_ptr()
        c.idx, c.idx_stride = self.idx.data_ptr(), self.idx.shape[1]
        c.nvalid, c.nvalid_stride = self.nvalid.data_ptr(), self.nvalid.shape[1]
        c.arena, c.arena_stride = arena.data_ptr(), arena_stride
        c.grad = self.grad.data_ptr()
        c.grad_stride = layout.trainable_numel if grad is not None else 0
        c.shadow = self.shadow.data_ptr()
        c.shadow_stride = layout.numel if arena_stride else 0
        c.w2t = self.w2t.data_ptr()
        c.w2t_stride = self.w2t.shape[1] if arena_stride else 0
        c.n_arena = layout.numel
        for field, key in (
            ("off_c1w", "conv1.weight"), ("off_c1b", "conv1.bias"),
            ("off_c2w", "conv2.weight"), ("off_c2b", "conv2.bias"),
            ("off_f1w", "fc1.weight"), ("off_f1b", "fc1.bias"),
            ("off_f2w", "fc2.weight"), ("off_f2b", "fc2.bias"),
        ):
            setattr(c, field, layout[key].offset)
        c.a1, c.p2, c.pmask = self.a1.data_ptr(), self.p2.data_ptr(), self.pmask.data_ptr()
        c.fc1_slabs, c.dh1, c.dp2 = (
            self.fc1_slabs.data_ptr(), self.dh1.data_ptr(), self.dp2.data_ptr())
        c.a1mask = self.a1mask.data_ptr()
        c.conv1_part, c.conv2_part, c.head_part = (
            self.conv1_part.data_ptr(), self.conv2_part.data_ptr(), self.head_part.data_ptr())
        c.c2b_part = self.c2b_part.data_ptr()
        # the SGD update of fc1.weight is applied by fc1_wgrad itself (DP-SGD: in the
        # clipped pass, noise included; the other parameters by finalize)
        # (micro-batches: the gradients are added up in the gradient arena first)
        c.fuse_sgd = 1 if (train and fuse_sgd and micro == 1) else 0
        c.micro = micro
        # The three backward kernels that only need what fc1_dgrad left behind share one
        # launch of four-wave workgroups, two of which fit a CU (same bits as the separate
        # launches).  Measured against the pair launch + conv2_wgrad, alternating runs on one
        # box: 89.8 against 79.4 rounds/s with one client on the GPU, 82.4 / 71.6 with two,
        # 62.8 / 58.1 with four, 44.9 / 42.1 with eight.  (With conv2_wgrad's six-wave body
        # the launch had 384-thread workgroups, one per CU, and only paid for one client.)
        # NANOFED_MNIST_TRIPLE=0|1 overrides (measurements).
        triple = os.environ.get("NANOFED_MNIST_TRIPLE")
        c.bwd_triple = int(triple == "1") if triple in ("0", "1") else 1
        c.ntotal = self.ntotal.data_ptr() if (train and (micro > 1 or cfg is not None
                                               and cfg.batch_size < BATCH)) else 0
        c.dp_normsq, c.dp_rowscale = self.dp_normsq.data_ptr(), self.dp_rowscale.data_ptr()
        c.dp_c2b = self.dp_c2b.data_ptr()
        if dp is not None and train:
            c.dp_pass = 2
            c.dp_clip = dp.max_gradient_norm
            c.dp_noise_std = dp.noise_multiplier * dp.max_gradient_norm
        c.metrics = self.metrics.data_ptr()
        c.eval_out = 0 if train else self.eval_out.data_ptr()
        c.step_counter = self.step_counter.data_ptr()
        c.seed = seed & (2**64 - 1)
        c.lr = cfg.learning_rate if cfg else 0.0
        c.wd = cfg.weight_decay if cfg else 0.0
        c.mean, c.inv_std = mean, 1.0 / std
        # model_kwargs={"dropout": False} builds the network without dropout (numerics
        # checks against fp32 with nothing random in the step)
        dropout = bool(cfg.model_kwargs.get("dropout", True)) if cfg is not None else True
        c.p_drop1, c.p_drop2 = (0.25, 0.5) if dropout else (0.0, 0.0)
        self.ctx = c

    def stream(self) -> c_void_p:
        return library.stream_ptr(self.device)

    def run_stage(self, name: str, idx_off: int = 0, step: int = 0) -> None:
        library.check(
            _lib().nf_mnist_stage(ctypes.byref(self.ctx), STAGES[name], idx_off, step,
                                  self.stream()),
            f"mnist stage {name}",
        )

    def pack(self) -> None:
        library.check(_lib().nf_mnist_pack(ctypes.byref(self.ctx), self.stream()), "mnist pack")

    def train_steps(self, first: int, count: int) -> None:
        library.check(
            _lib().nf_mnist_train_steps(ctypes.byref(self.ctx), first, count, self.stream()),
            "mnist train steps",
        )

    def eval_step(self, idx_off: int, step: int) -> None:
        library.check(
            _lib().nf_mnist_eval_step(ctypes.byref(self.ctx), idx_off, step, self.stream()),
            "mnist eval step",
        )


class MnistCnnNativeTrainer(GroupTrainer):
    make_layout = staticmethod(make_layout)
    # epochs spent choosing the group streams (NANOFED_MNIST_STREAM_TRIALS: measurements)
    STREAM_TRIALS = max(1, int(os.environ.get("NANOFED_MNIST_STREAM_TRIALS", "4")))
    # ... raised to this many when the two fastest sets of the first draws differ by more
    # than 5 % (no plateau found yet: the fastest may be a lucky draw, or a better one exists)
    STREAM_TRIALS_MAX = max(STREAM_TRIALS, int(os.environ.get("NANOFED_MNIST_STREAM_TRIALS_MAX", "8")))
    # the set that won, per (device, number of streams), for the life of the process: a
    # second trainer on the same device starts on it without drawing again
    _WINNERS: dict[tuple[int, int], tuple[list, dict]] = {}
    HEAD_STEPS = 8     # steps in the short first graph of a group's epoch

    def __init__(
        self, cfg: EngineConfig, pool: ArenaPool, shards: DeviceShards, data: FederatedData
    ) -> None:
        if cfg.model != "mnist_cnn":
            raise DeviceError("the native trainer implements the MNIST CNN only")
        if not supports(cfg):
            raise DeviceError(
                f"the native MNIST trainer covers batch sizes 1..{MAX_MICRO * BATCH} "
                f"(DP-SGD: {BATCH}) of the reference architecture"
            )
        self.micro = micro_batches(cfg.batch_size)
        if not pool.layout.perms:
            raise DeviceError("the arena pool must use MnistCnnNativeTrainer.make_layout")
        self.cfg, self.pool, self.shards = cfg, pool, shards
        self.device = pool.data.device
        G = len(pool)
        self.max_steps = shards.max_steps()
        mean, std = float(data.mean[0]), float(data.std[0])
        self.buf = NativeBuffers(
            pool.layout, G, self.device, shards.inputs, shards.labels, pool.data,
            pool.layout.numel, pool.grad, self.max_steps, mean, std, cfg,
            client_base=shards.client_ids[0] if shards.client_ids else 0,
            train=True, seed=cfg.seed, dp=cfg.dp, micro=self.micro,
        )
        self._accountants = None
        if cfg.dp is not None:
            if cfg.dp.mode != "per_sample":
                raise DeviceError("the native DP path clips per sample (DPConfig.mode='per_sample')")
            from nanofed_amd.privacy import PrivacyConfig
            from nanofed_amd.privacy.accountant.rdp import SubsampledRDPAccountant

            pc = PrivacyConfig(
                epsilon=min(max(cfg.dp.epsilon, 0.01), 10.0), delta=cfg.dp.delta,
                max_gradient_norm=cfg.dp.max_gradient_norm,
                noise_multiplier=cfg.dp.noise_multiplier,
            )
            self._accountants = [
                SubsampledRDPAccountant(pc, dataset_size=max(n, 1)) for n in shards.sizes
            ]
        # Tables per client: table step t = s * micro + j is micro-batch j of batch s and
        # holds samples [s * batch + j * 64, ...) of the epoch's permutation.
        B, m, bs = BATCH, self.micro, cfg.batch_size
        T = self.max_steps * m
        nvalid = torch.zeros(G, T, dtype=torch.int32)
        ntotal = torch.zeros(G, T, dtype=torch.int32)
        t = torch.arange(T)
        lane = torch.arange(B)
        # position in the permutation of row `lane` of table step t, and whether the
        # batch reaches that far
        pos = ((t // m) * bs + (t % m) * B).unsqueeze(1) + lane.unsqueeze(0)        # [T, B]
        in_batch = ((t % m) * B).unsqueeze(1) + lane.unsqueeze(0) < bs
        self._gather_pos, self._gather_ok = [], []
        for k in range(G):
            n = shards.sizes[k]
            ok = in_batch & (pos < n)
            if shards.drop_last:
                full = ((t // m) + 1) * bs <= n
                ok = ok & full.unsqueeze(1)
            nvalid[k] = ok.sum(dim=1).to(torch.int32)
            per_batch = nvalid[k].view(self.max_steps, m).sum(dim=1)
            ntotal[k] = per_batch.repeat_interleave(m)
            # (numpy: the tables are built on the host every epoch, and numpy's gathers do
            # not go through a thread pool that may have to wake up first)
            self._gather_pos.append(torch.where(ok, pos, torch.zeros_like(pos)).reshape(-1).numpy())
            self._gather_ok.append(ok.reshape(-1).numpy())
        self.buf.nvalid.copy_(nvalid)
        self.buf.ntotal.copy_(ntotal)
        # unequal shards: the kernels skip a client's workgroups once it has no batch left
        self.buf.ctx.ragged = int(bool((nvalid == 0).any()))
        self.metrics = torch.zeros(G, METRIC_COLUMNS, dtype=torch.float32, device=self.device)
        # Client groups: with two or more clients on the GPU the step runs as up to
        # four launch chains of ceil(G / 4) clients each, on concurrent streams (they
        # share no data).  Measured per 8-client step inside 118-step epoch graphs: 115 us
        # for 4 x 2 against 127 us for one chain of 8.  The groups are views of
        # self.buf (same batches, activations, counters).
        self._groups: list[NativeBuffers] = [self.buf]
        self._group_streams: list[torch.cuda.Stream] = []
        sizes = self._group_sizes(G) if cfg.client_streams else [G]
        if len(sizes) > 1:
            self._groups = []
            first = 0
            for gs in sizes:
                grp = NativeBuffers(
                    pool.layout, gs, self.device, self.buf.images, self.buf.labels,
                    pool.data[first:], pool.layout.numel,
                    pool.grad[first:] if pool.grad is not None else None, self.max_steps,
                    mean, std, cfg, client_base=shards.client_ids[first] if shards.client_ids else first,
                    train=True, seed=cfg.seed, dp=cfg.dp, parent=self.buf, first=first,
                    micro=self.micro,
                )
                grp.ctx.ragged = int(bool((nvalid[first : first + gs] == 0).any()))
                self._groups.append(grp)
                first += gs
            # (the streams are created at the first epoch, see _run_epoch)
        self.stream_report: dict = {}
        self._graphs: list[list[torch.cuda.CUDAGraph]] | None = None
        self._chunks: list[tuple[int, int]] = []
        self._stream_trials: list[tuple[float, list[torch.cuda.Stream]]] = []
        self._trial_target = self.STREAM_TRIALS
        # the next epoch's index table, built on the host while the current epoch runs:
        # a ring of page-locked tables (allocated once: page-locking memory can take tens
        # of milliseconds), each guarded by the event of the last copy that read it
        self._idx_ring = [torch.empty_like(self.buf.idx, device="cpu").pin_memory() for _ in range(3)]
        self._idx_ring_events: list[torch.cuda.Event | None] = [None] * len(self._idx_ring)
        self._idx_ring_next = 0
        self._idx_host: int | None = None  # ring slot that holds the table of _idx_key
        self._idx_key: tuple[int, int] | None = None
        self._eval: NativeBuffers | None = None
        self._eval_data = (data.eval_inputs, data.eval_labels, mean, std)
        self._sizes = torch.tensor(
            [float(s) for s in shards.sizes], dtype=torch.float32, device=self.device
        )
        # optimisation steps a client takes per local epoch: its step counter (which
        # numbers the dropout and noise streams) restarts every round from
        # round * local_epochs * steps, so a rolled-back or resumed run draws the
        # same streams as an uninterrupted one and nothing has to be checkpointed
        self._steps_per_epoch = (nvalid > 0).sum(dim=1).to(dtype=torch.int64, device=self.device)

    @staticmethod
    def _group_sizes(G: int) -> list[int]:
        """Up to four chains (a process has four hardware queues; more chains than queues
        run slower, measured), i.e. groups of ceil(G / 4) clients.  Measured: 64 clients
        5.0 M samples/s as 4 x 16 against 4.3 M as 32 pairs; 4 clients 51.1 rounds/s as
        4 x 1 against 49.5 as 2 x 2 and 43.0 as one chain; 2 clients 60.3 as 2 x 1
        against 58.6 as one chain.  ``NANOFED_MNIST_GROUP`` overrides the group size
        (measurements)."""
        if "NANOFED_MNIST_GROUP" in os.environ:
            size = max(1, min(int(os.environ["NANOFED_MNIST_GROUP"]), G))
        else:
            size = max(1, -(-G // 4))
        return [size] * (G // size) + ([G % size] if G % size else [])

    # -- batches ----------------------------------------------------------
    def _host_indices(self, round_idx: int, epoch: int) -> int:
        """Build the epoch's index table [clients, table steps * 64] - the clients' shuffles
        (drawn on the host) gathered into batch order - in the next page-locked table of the
        ring; returns its slot."""
        slot = self._idx_ring_next
        self._idx_ring_next = (slot + 1) % len(self._idx_ring)
        last_copy = self._idx_ring_events[slot]
        if last_copy is not None:
            last_copy.synchronize()  # two tables ago: has run long since
        table = self._idx_ring[slot].numpy()
        for k in range(len(self.pool)):
            perm = self.shards.host_permutation(k, round_idx, epoch).numpy()
            # rows beyond the batch or the shard repeat the first sample (zero loss weight)
            row = perm[self._gather_pos[k]]
            np.copyto(table[k], np.where(self._gather_ok[k], row, perm[0]), casting="unsafe")
        return slot

    def _copy_indices(self, slot: int) -> None:
        self.buf.idx.copy_(self._idx_ring[slot], non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        self._idx_ring_events[slot] = done

    def _fill_indices(self, round_idx: int, epoch: int) -> None:
        self._copy_indices(self._host_indices(round_idx, epoch))

    # An epoch's shuffles used to be a few dozen tiny sort launches per client on a side
    # stream; a process has four hardware queues, the side stream shared one with a
    # training chain, and the training stream waited 0.3-0.4 ms per epoch for them.  The
    # host, idle while an epoch's launches run, draws them now: the table of the next
    # epoch is built right after the current epoch has been queued, and the training
    # stream pays one 240 KB copy.
    def _prefetch_indices(self, round_idx: int, epoch: int) -> None:
        self._idx_host = self._host_indices(round_idx, epoch)
        self._idx_key = (round_idx, epoch)

    def _take_indices(self, round_idx: int, epoch: int) -> None:
        if self._idx_key != (round_idx, epoch):
            self._prefetch_indices(round_idx, epoch)
        self._copy_indices(self._idx_host)
        self._idx_host, self._idx_key = None, None

    def _ensure_graph(self, round_idx: int) -> None:
        """Record one local epoch (every step of every client of a group) in a
        hipGraph per group.  The eager warm-up that precedes the capture trains for
        real, so the parameters and the dropout counters are restored afterwards."""
        if not self.cfg.graph_capture or self._graphs is not None:
            return
        saved = self.pool.data.clone()
        counter = self.buf.step_counter.clone()
        self.buf.pack()
        self._fill_indices(round_idx, 0)
        # The host needs about 0.3 ms to enqueue a 118-step graph, so the groups' epochs
        # would start staggered by that much (and a single chain would start that late
        # after the host synchronisation that ends a round).  Each group's epoch is
        # therefore two graphs, a short head and the rest, launched head-first across
        # the groups (all chains are running while the long graphs go out).
        head = self.HEAD_STEPS if self.max_steps > self.HEAD_STEPS else 0
        self._chunks = [(0, head), (head, self.max_steps - head)] if head else [(0, self.max_steps)]
        graphs = []
        for grp in self._groups:
            grp.train_steps(0, self.max_steps)
            torch.cuda.synchronize(self.device)
            per_chunk = []
            for first, count in self._chunks:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    grp.train_steps(first, count)
                per_chunk.append(graph)
            graphs.append(per_chunk)
        self._graphs = graphs
        # choose the group streams now (timed epochs on candidate sets, see
        # _run_epoch), before the state is put back: no round pays for it
        while len(self._groups) > 1 and not self._group_streams:
            self._run_epoch()
        self.pool.data.copy_(saved)
        self.buf.step_counter.copy_(counter)

    def _run_epoch(self) -> None:
        if len(self._groups) == 1:
            if self._graphs is not None:
                for graph in self._graphs[0]:
                    graph.replay()
            else:
                self.buf.train_steps(0, self.max_steps)
            return
        main = torch.cuda.current_stream(self.device)
        streams = self._group_streams
        calibrating = not streams
        if calibrating:
            want = min(len(self._groups), int(os.environ.get("NANOFED_MNIST_STREAMS", "4")))
            key = (self.device.index or 0, want)
            if key in self._WINNERS and os.environ.get("NANOFED_MNIST_STREAM_MEMO", "1") != "0":
                streams, report = self._WINNERS[key]
                self._group_streams = streams
                self.stream_report = dict(report, groups=len(self._groups), reused=True)
                calibrating = False
        if calibrating:
            # Which hardware queue a stream lands on is the runtime's business, and
            # stream sets differ a lot (21 against 14 ms per epoch measured: streams
            # that share a queue run their chains one after the other).  The first
            # epochs therefore each run on a newly created set and are timed; the
            # fastest set is kept (with graphs: during the set-up of the first round,
            # whose training is rolled back anyway; eager: the first epochs of training).
            from nanofed_amd.utils.streams import make_streams

            streams = make_streams(want, self.device)
            t0 = torch.cuda.Event(enable_timing=True)
            t1 = torch.cuda.Event(enable_timing=True)
            t0.record(main)
        for st in streams:
            st.wait_stream(main)
        if self._graphs is not None:
            for c in range(len(self._chunks)):
                for i in range(len(self._groups)):
                    with torch.cuda.stream(streams[i % len(streams)]):
                        self._graphs[i][c].replay()
        else:
            for i, grp in enumerate(self._groups):
                with torch.cuda.stream(streams[i % len(streams)]):
                    grp.train_steps(0, self.max_steps)
        for st in streams:
            main.wait_stream(st)
        if calibrating:
            t1.record(main)
            t1.synchronize()
            self._stream_trials.append((t0.elapsed_time(t1), streams))
            self._settle_streams(key)

    def _settle_streams(self, key: tuple[int, int]) -> None:
        """After a timed draw: keep drawing, or keep the fastest set."""
        trials = self._stream_trials
        if len(trials) < self._trial_target:
            return
        times = sorted(t for t, _ in trials)
        if (len(times) >= 2 and self._trial_target < self.STREAM_TRIALS_MAX
                and times[1] > 1.05 * times[0]):
            self._trial_target = self.STREAM_TRIALS_MAX
            return
        best = min(range(len(trials)), key=lambda i: trials[i][0])
        self._group_streams = trials[best][1]
        # what was tried and what was kept goes into the round report; the sets that
        # lost are handed back to the runtime
        self.stream_report = {
            "trial_epoch_ms": [round(t, 3) for t, _ in trials],
            "chosen_trial": best,
            "groups": len(self._groups),
            "streams": len(self._group_streams),
            "reused": False,
        }
        from nanofed_amd.utils.streams import destroy_streams

        torch.cuda.synchronize(self.device)
        for i, (_, losers) in enumerate(trials):
            if i != best:
                destroy_streams(losers)
        self._stream_trials = [(t, []) if i != best else (t, st) for i, (t, st) in enumerate(trials)]
        self._WINNERS[key] = (self._group_streams, dict(self.stream_report))

    def train_round(self, round_idx: int) -> torch.Tensor:
        cfg = self.cfg
        self._ensure_graph(round_idx)
        self.buf.pack()
        torch.mul(self._steps_per_epoch, round_idx * cfg.local_epochs, out=self.buf.step_counter)
        for epoch in range(cfg.local_epochs):
            self._take_indices(round_idx, epoch)
            self.buf.metrics.zero_()
            self._run_epoch()
            nxt = (round_idx, epoch + 1) if epoch + 1 < cfg.local_epochs else (round_idx + 1, 0)
            self._prefetch_indices(*nxt)
        m = self.buf.metrics
        batches = m[:, 2].clamp(min=1.0)
        out = self.metrics
        out.zero_()
        out[:, M_SAMPLES] = self._sizes
        out[:, M_LOSS] = m[:, 0] / batches
        out[:, M_ACC] = m[:, 1] / batches
        out[:, M_VALID] = 1.0
        if self._accountants is not None:
            # privacy spent is a pure function of the number of noisy steps taken
            # since round 0: one closed-form evaluation per client and round
            eps = [
                acc.epsilon_after(
                    (round_idx + 1) * cfg.local_epochs * self.shards.steps_per_epoch(k),
                    cfg.dp.noise_multiplier, BATCH,
                )
                for k, acc in enumerate(self._accountants)
            ]
            # from page-locked memory: the copy does not make the host wait for the epochs
            # queued before it
            out[:, M_EPSILON].copy_(torch.tensor(eps, dtype=torch.float32).pin_memory(),
                                    non_blocking=True)
        return out

    # -- held-out evaluation ----------------------------------------------
    @torch.no_grad()
    def evaluate(self, flat: torch.Tensor) -> dict[str, float]:
        ex, ey, mean, std = self._eval_data
        if ex is None or len(ex) == 0:
            return {}
        B = BATCH
        n = len(ex)
        nb = (n + B - 1) // B
        GE = min(8, nb)
        steps = (nb + GE - 1) // GE
        if self._eval is None:
            ev = NativeBuffers(
                self.pool.layout, GE, self.device, ex.to(self.device), ey.to(self.device),
                flat, 0, None, steps, mean, std, None, train=False,
            )
            idx = torch.zeros(GE, steps * B, dtype=torch.int32)
            nvalid = torch.zeros(GE, steps, dtype=torch.int32)
            for batch in range(nb):
                g, s = batch % GE, batch // GE
                lo, hi = batch * B, min(n, batch * B + B)
                idx[g, s * B : s * B + (hi - lo)] = torch.arange(lo, hi, dtype=torch.int32)
                nvalid[g, s] = hi - lo
            ev.idx.copy_(idx)
            ev.nvalid.copy_(nvalid)
            ev.ctx.ragged = int(bool((nvalid == 0).any()))
            self._eval = ev
            self._eval_active = (nvalid > 0).t().reshape(steps, GE, 1, 1).to(
                device=self.device, dtype=torch.float64)
        ev = self._eval
        if ev.arena.data_ptr() != flat.data_ptr():
            ev.arena = flat
            ev.ctx.arena = flat.data_ptr()
        ev.pack()
        # every workgroup leaves its loss sum / hit count of the step in its own slot;
        # the slots of the clients that had a batch are added here in fp64, in a
        # fixed order, so the result does not depend on timing
        total = torch.zeros(2, dtype=torch.float64, device=self.device)
        for s in range(steps):
            ev.eval_step(s * B, s)
            total += (ev.eval_out.double() * self._eval_active[s]).sum(dim=(0, 1))
        loss_sum, correct = total.tolist()
        return {"loss": loss_sum / n, "accuracy": correct / n, "samples": float(n)}
