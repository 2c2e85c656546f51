"""Local trainers of the device plane: each trains the G simulated clients
that share one GPU for one federated round.

``TorchGroupTrainer`` is the model-agnostic path.  It keeps the reference's
training semantics (nanofed/trainer/base.py:115-198: per batch zero_grad,
forward, cross-entropy, backward, SGD step; metrics of the last local epoch)
but

* every client's parameters live in one arena pool; in bf16 its weight
  matrices are bf16 leaves over a bf16 copy of the arena, autograd's gradient
  tensors are consumed as they are, and the optimizer is one multi-tensor HIP
  kernel per step (fp32 / DP runs: arena-view parameters, a flat gradient
  buffer and one flat SGD kernel that also clears the gradients);
* batches are index gathers from the HBM-resident shard, not H2D copies;
* loss and accuracy are accumulated in a device tensor - nothing is read
  back during the round (the reference synchronises twice per batch,
  base.py:154,159);
* a client's whole step is one hipGraph replay, and the clients of a GPU run
  on concurrent streams (image models).

Forward/backward are PyTorch-ROCm ops under bf16 autocast.  The MNIST CNN has
a fully hand-written path on top of this interface
(``nanofed_amd.engine.mnist_native``).
"""

from __future__ import annotations

import os
from typing import Callable

import torch
import torch.nn as nn
import torch.nn.functional as F

from nanofed_amd import ops
from nanofed_amd.core.exceptions import DeviceError
from nanofed_amd.runtime.arena import ArenaPool
from nanofed_amd.utils.streams import side_stream

from .config import EngineConfig
from .data import DeviceShards

# columns of the per-client metrics tensor
M_SAMPLES, M_LOSS, M_ACC, M_VALID, M_UPDATE_SQNORM, M_EPSILON = range(6)
METRIC_COLUMNS = 6


def classification_loss(output: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """Cross-entropy over the last dimension (images: [B,C]; LM: [B,S,V])."""
    if output.dim() > 2:
        output = output.reshape(-1, output.shape[-1])
        target = target.reshape(-1)
    return F.cross_entropy(output.float(), target)


def count_correct(output: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    pred = output.argmax(dim=-1)
    return (pred == target).float().mean()


class GroupTrainer:
    """Interface of a per-GPU trainer."""

    def train_round(self, round_idx: int) -> torch.Tensor:  # [G, METRIC_COLUMNS]
        raise NotImplementedError

    def evaluate(self, flat: torch.Tensor) -> dict[str, float]:
        raise NotImplementedError


class TorchGroupTrainer(GroupTrainer):
    def __init__(
        self,
        cfg: EngineConfig,
        pool: ArenaPool,
        shards: DeviceShards,
        model_factory: Callable[[], nn.Module],
        eval_inputs: torch.Tensor | None = None,
        eval_labels: torch.Tensor | None = None,
    ) -> None:
        self.cfg = cfg
        self.pool = pool
        self.shards = shards
        self.device = pool.data.device
        self.modules = []
        for k in range(len(pool)):
            m = model_factory().to(self.device)
            m.train()
            self.modules.append(pool[k].bind(m))
        self._eval_module = model_factory().to(self.device)
        self._eval_inputs = eval_inputs
        self._eval_labels = eval_labels
        self.metrics = torch.zeros(
            len(pool), METRIC_COLUMNS, dtype=torch.float32, device=self.device
        )
        self._acc = torch.zeros(len(pool), 3, dtype=torch.float32, device=self.device)
        self._amp = cfg.dtype == "bf16" and self.device.type == "cuda"
        self._dp_step = 0
        self._accountants = None
        if cfg.dp is not None:
            if cfg.dp.mode not in ("per_sample", "batch"):
                raise DeviceError(f"unknown DPConfig.mode {cfg.dp.mode!r}")
            if cfg.dp.mode == "per_sample" and any(
                isinstance(m, nn.modules.batchnorm._BatchNorm) for m in self.modules[0].modules()
            ):
                raise DeviceError(
                    "per-sample clipping needs independent examples; a model with BatchNorm "
                    "takes DPConfig(mode='batch')"
                )
        if cfg.dp is not None and cfg.dp.noise_multiplier > 0:
            # (without noise there is nothing to account for: clipping alone is not private)
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
        # whole-step hipGraphs (forward + backward + SGD of one client): DP draws
        # its noise offset on the host, so it stays eager
        self._use_graph = (
            cfg.graph_capture and self.device.type == "cuda" and cfg.dp is None
        )
        self._graphs: dict[int, tuple[torch.cuda.CUDAGraph, torch.Tensor]] = {}
        self._channels_last = self.device.type == "cuda" and shards.kind == "image"
        # bf16 training without per-step casts and gradient accumulation kernels:
        # matrices and convolution weights are bf16 leaves (views of a bf16 copy of
        # the arena that the optimizer refreshes), autograd hands out each gradient
        # as its own tensor, and one multi-tensor kernel applies them to the fp32
        # arena.  DP needs the flat gradient buffer and keeps the accumulate path.
        self._direct = self._amp and cfg.dp is None
        self._shadow: torch.Tensor | None = None
        self._leaves: list[list[tuple[nn.Parameter, int]]] = []
        if self._direct:
            self._bind_shadow()
        # BatchNorm layers on the fused kernels leave their step counter to
        # train_round, which bumps it once per epoch instead of once per step
        self._bn_layers: list[list[nn.Module]] = [[] for _ in self.modules]
        counted: list[str] = []  # their step counters, kept in the arena rows (below)
        if self._channels_last and self._amp:
            from nanofed_amd.ops import bn_ops

            for k, module in enumerate(self.modules):
                for name, m in module.named_modules():
                    if isinstance(m, nn.BatchNorm2d) and bn_ops.channels_supported(m.num_features):
                        m.defer_batch_count = True
                        self._bn_layers[k].append(m)
                        if k == 0 and f"{name}.num_batches_tracked" in pool.layout:
                            counted.append(f"{name}.num_batches_tracked")
        # Step counters of those layers live in the client's arena row only: the row starts
        # every round from the global model (also after a rollback), the trainer adds the
        # epoch's steps to it, and the module's own counter is never consulted.  Other
        # integer buffers travel module <-> arena around the round.
        self._counted = frozenset(counted)
        self._count_idx = torch.tensor([pool.layout[n].offset for n in counted], dtype=torch.long,
                                       device=self.device)
        self._other_ints = any(
            not b.dtype.is_floating_point and n not in self._counted and n in pool.layout
            for n, b in self.modules[0].named_buffers()
        ) if self.modules else False
        # Client streams (default priority) for the convolutional models.  Token
        # models stay on one stream: their large GEMMs may run as stream-K kernels,
        # whose workgroups wait for each other through memory - two of those
        # sharing the chip can wait forever (seen with the GPT-2 trainer).
        self._streams: list[torch.cuda.Stream] | None = None
        if (cfg.client_streams and self.device.type == "cuda" and len(pool) > 1
                and shards.kind == "image"):
            # NANOFED_CLIENT_STREAMS overrides the configured count for experiments
            ns = int(os.environ.get("NANOFED_CLIENT_STREAMS", cfg.client_stream_count))
            ns = max(1, min(len(pool), ns))
            base = [side_stream(self.device) for _ in range(ns)]
            self._streams = [base[k % ns] for k in range(len(pool))]
        # The convolution library times its kernels per shape once (during the
        # eager warm-up steps) instead of taking its untuned default.  The flag is
        # process-global in PyTorch, so it is switched on only around this trainer's
        # own work (train_round / evaluate) and put back afterwards.
        self._autotune = self._channels_last

    def _bind_shadow(self) -> None:
        layout = self.pool.layout
        self._shadow = torch.empty(
            len(self.pool), layout.trainable_numel, dtype=torch.bfloat16, device=self.device
        )
        for k, module in enumerate(self.modules):
            leaves = []
            for name, p in module.named_parameters():
                seg = layout[name]
                p.grad = None
                if p.dim() >= 2:
                    p.data = layout.view_of(self._shadow[k], seg)
                leaves.append((p, seg.offset))
            self._leaves.append(leaves)
        self._refresh_shadow()

    def _refresh_shadow(self, k: int | None = None) -> None:
        """bf16 copy of the (new) fp32 parameters: after a broadcast of the global
        model or a restore of the arena."""
        n = self.pool.layout.trainable_numel
        if k is None:
            self._shadow.copy_(self.pool.data[:, :n])
        else:
            self._shadow[k].copy_(self.pool[k].data[:n])

    def _apply_gradients(self, k: int) -> None:
        cfg = self.cfg
        grads, offsets = [], []
        for p, off in self._leaves[k]:
            g = p.grad
            if g is None:
                continue
            if g.stride() != p.stride():  # bring it into the arena's element order
                g = torch.empty_strided(p.shape, p.stride(), dtype=g.dtype, device=g.device).copy_(g)
            grads.append(g)
            offsets.append(off)
            p.grad = None
        ops.sgd_multi(
            self.pool[k].trainable, grads, offsets, cfg.learning_rate, cfg.weight_decay,
            bf16_shadow=self._shadow[k],
        )

    # -- one optimisation step of client k --------------------------------
    def _step(self, k: int, rows: torch.Tensor) -> None:
        """Full batches replay the client's captured step; the ragged last
        batch of an epoch runs eagerly."""
        if self._use_graph and rows.numel() == self.cfg.batch_size:
            graph, static_rows = self._graph_for(k)
            static_rows.copy_(rows)
            graph.replay()
        else:
            self._eager_step(k, rows)

    def _graph_for(self, k: int) -> tuple[torch.cuda.CUDAGraph, torch.Tensor]:
        """Capture client k's optimisation step once.  The warm-up iterations
        (library autotuning, lazy allocations) really train, so the client's
        parameters, float buffers and counters are restored afterwards."""
        if k in self._graphs:
            return self._graphs[k]
        arena = self.pool[k]
        keep = arena.data.clone()
        keep_acc = self._acc[k].clone()
        int_bufs = [b.clone() for b in self.modules[k].buffers() if not b.dtype.is_floating_point]
        static_rows = (
            torch.arange(self.cfg.batch_size, device=self.device) + int(self.shards.offsets[k])
        )
        side = side_stream(self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                self._eager_step(k, static_rows)
            # the ragged last batch of an epoch is a shape of its own: let the
            # convolution library tune it now, not in the middle of a round while
            # the other clients' graphs replay
            tail = self.shards.sizes[k] % self.cfg.batch_size
            if tail and not self.shards.drop_last:
                self._eager_step(k, static_rows[:tail])
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self._eager_step(k, static_rows)
        arena.data.copy_(keep)
        if arena.grad is not None:
            arena.grad.zero_()
        if self._direct:
            self._refresh_shadow(k)
        self._acc[k].copy_(keep_acc)
        live = [b for b in self.modules[k].buffers() if not b.dtype.is_floating_point]
        for b, saved in zip(live, int_bufs):
            b.copy_(saved)
        self._graphs[k] = (graph, static_rows)
        return self._graphs[k]

    def _eager_step(self, k: int, rows: torch.Tensor) -> None:
        cfg = self.cfg
        module = self.modules[k]
        arena = self.pool[k]
        x, y = self.shards.batch(rows)
        if self._channels_last:
            x = x.contiguous(memory_format=torch.channels_last)
        if cfg.dp is not None and cfg.dp.mode == "per_sample":
            loss, out = self._per_sample_step(k, x, y)
        else:
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self._amp):
                out = module(x)
                loss = classification_loss(out, y)
            loss.backward()
            if cfg.dp is not None:
                self._privatise(k, int(rows.numel()))
        if self._direct:
            self._apply_gradients(k)
        else:
            ops.sgd_step(
                arena.trainable, arena.grad, cfg.learning_rate, cfg.weight_decay,
                zero_grad=True,
            )
        acc = self._acc[k]
        acc[0] += loss.detach()
        acc[1] += count_correct(out.detach(), y)
        acc[2] += 1.0

    def _per_sample_step(self, k: int, x: torch.Tensor, y: torch.Tensor):
        """DP-SGD gradient of one batch (Abadi et al.): every example's gradient is
        clipped to ``max_gradient_norm``, the clipped gradients are averaged and
        Gaussian noise of standard deviation ``noise_multiplier * max_gradient_norm
        / batch`` is added.  The per-example gradients come from ``torch.func``
        (vmap over grad of the functional model) - the portable counterpart of the
        native trainer's per-sample kernels.  Leaves the result in the client's
        flat gradient buffer; returns (mean loss, outputs)."""
        from torch.func import functional_call, grad, vmap

        dp = self.cfg.dp
        module = self.modules[k]
        arena = self.pool[k]
        params = {n: p.detach() for n, p in module.named_parameters()}
        buffers = {n: b for n, b in module.named_buffers()}

        def loss_one(p, xi, yi):
            out = functional_call(module, (p, buffers), (xi.unsqueeze(0),))
            return classification_loss(out, yi.unsqueeze(0)), (out[0], )

        grads, (outs,) = vmap(
            grad(loss_one, has_aux=True), in_dims=(None, 0, 0), randomness="different"
        )(params, x, y)
        B = x.shape[0]
        sq = torch.zeros(B, dtype=torch.float32, device=x.device)
        for g in grads.values():
            sq += g.reshape(B, -1).float().square().sum(dim=1)
        factor = (dp.max_gradient_norm / sq.sqrt().clamp(min=1e-12)).clamp(max=1.0)
        gviews = arena.grad_views()
        for name, g in grads.items():
            mean = (g.float() * factor.view(B, *([1] * (g.dim() - 1)))).sum(dim=0) / B
            gviews[name].copy_(mean)
        cid = self.shards.client_ids[k]
        self._dp_step += 1
        std = dp.noise_multiplier * dp.max_gradient_norm / B
        if std > 0:
            # clip bound far above any norm: this call only adds the noise
            ops.clip_noise(arena.grad, 1e30, std, seed=self.cfg.seed * 2654435761 + cid,
                           offset=self._dp_step, out=arena.grad)
        with torch.no_grad():
            loss = classification_loss(outs, y)
        return loss, outs

    def _privatise(self, k: int, batch: int) -> None:
        """Batch-level clip + Gaussian noise on the gradient arena - the
        reference's DP step (nanofed/trainer/private.py:54-86) as two fused
        passes over one buffer."""
        dp = self.cfg.dp
        g = self.pool[k].grad
        cid = self.shards.client_ids[k]
        self._dp_step += 1
        ops.clip_noise(
            g, dp.max_gradient_norm, dp.noise_multiplier * dp.max_gradient_norm,
            seed=self.cfg.seed * 2654435761 + cid, offset=self._dp_step, out=g,
        )

    def train_round(self, round_idx: int) -> torch.Tensor:
        # PyTorch's dropout draws from the process-wide generators.  The round runs
        # on a fork of them, seeded from (run seed, round, first client), so that its
        # randomness is a function of the round number - a repeated or resumed round,
        # or a second engine in the same process, draws the same masks - and the
        # caller's generators are left as they were.
        devices = [self.device] if self.device.type == "cuda" else []
        first = self.shards.client_ids[0] if self.shards.client_ids else 0
        with torch.random.fork_rng(devices=devices):
            torch.manual_seed(
                (self.cfg.seed * 1000003 + round_idx * 7919 + first * 104729 + 17) & 0x7FFFFFFFFFFFFFFF
            )
            if not self._autotune:
                return self._train_round(round_idx)
            previous = torch.backends.cudnn.benchmark
            torch.backends.cudnn.benchmark = True
            try:
                return self._train_round(round_idx)
            finally:
                torch.backends.cudnn.benchmark = previous

    def _train_round(self, round_idx: int) -> torch.Tensor:
        cfg = self.cfg
        G = len(self.pool)
        self.metrics.zero_()
        # the noise offsets of a round are a function of the round number (a
        # rolled-back round draws the same noise as the run that never failed)
        self._dp_step = round_idx * cfg.local_epochs * max(self.shards.max_steps(), 1) * max(G, 1)
        if self._direct:
            self._refresh_shadow()
        if self._other_ints:
            for k in range(G):
                self.pool[k].push_int_buffers(self.modules[k], skip=self._counted)
        if self._streams is not None and G > 1:
            self._train_clients_concurrently(round_idx)
        else:
            for k in range(G):
                # one client after the other: its dropout stream depends on its own
                # id only, so the result does not depend on how clients map to ranks
                cid = self.shards.client_ids[k]
                torch.manual_seed(
                    (cfg.seed * 1000003 + round_idx * 7919 + cid * 104729 + 17) & 0x7FFFFFFFFFFFFFFF
                )
                for epoch in range(cfg.local_epochs):
                    self._acc[k].zero_()
                    perm = self.shards.epoch_permutation(k, round_idx, epoch)
                    steps = self.shards.steps_per_epoch(k)
                    for s in range(steps):
                        self._step(k, perm[s * cfg.batch_size : (s + 1) * cfg.batch_size])
                    self._count_steps(k, steps)
        for k in range(G):
            batches = self._acc[k, 2].clamp(min=1.0)
            self.metrics[k, M_SAMPLES] = float(self.shards.sizes[k])
            self.metrics[k, M_LOSS] = self._acc[k, 0] / batches
            self.metrics[k, M_ACC] = self._acc[k, 1] / batches
            self.metrics[k, M_VALID] = 1.0
            if self._other_ints:
                self.pool[k].pull_int_buffers(self.modules[k], skip=self._counted)
        if self._accountants is not None:
            # privacy spent is a function of the number of noisy steps since round 0
            eps = [
                acc.epsilon_after(
                    (round_idx + 1) * cfg.local_epochs * self.shards.steps_per_epoch(k),
                    cfg.dp.noise_multiplier, cfg.batch_size,
                )
                for k, acc in enumerate(self._accountants)
            ]
            host = torch.tensor(eps, dtype=torch.float32)
            if self.device.type == "cuda":
                host = host.pin_memory()  # the copy does not wait for the queued steps
            self.metrics[:, M_EPSILON].copy_(host, non_blocking=True)
        return self.metrics

    def _count_steps(self, k: int, steps: int) -> None:
        """Add an epoch's optimisation steps to the BatchNorm step counters of client k."""
        if self._count_idx.numel():
            self.pool.data[k].index_add_(
                0, self._count_idx,
                torch.full((self._count_idx.numel(),), float(steps), device=self.device))

    def _train_clients_concurrently(self, round_idx: int) -> None:
        """Client k's steps go to stream k, enqueued step-major so that every
        stream has work: the clients are independent within a round, and one
        client's kernels (batch-64 layers) leave most of the chip idle."""
        cfg = self.cfg
        G = len(self.pool)
        main = torch.cuda.current_stream()
        if self._use_graph:
            for k in range(G):
                if self.shards.sizes[k] >= cfg.batch_size:
                    self._graph_for(k)  # capture on the main stream, one at a time
        steps = [self.shards.steps_per_epoch(k) for k in range(G)]
        perms = [
            [self.shards.epoch_permutation(k, round_idx, e) for e in range(cfg.local_epochs)]
            for k in range(G)
        ]
        for st in set(self._streams):
            st.wait_stream(main)

        for epoch in range(cfg.local_epochs):
            for k in range(G):
                with torch.cuda.stream(self._streams[k]):
                    self._acc[k].zero_()
            for s in range(max(steps)):
                for k in range(G):
                    if s < steps[k]:
                        with torch.cuda.stream(self._streams[k]):
                            lo = s * cfg.batch_size
                            self._step(k, perms[k][epoch][lo : lo + cfg.batch_size])
            for k in range(G):
                with torch.cuda.stream(self._streams[k]):
                    self._count_steps(k, steps[k])
        for st in set(self._streams):
            main.wait_stream(st)
        del perms  # released only after the main stream has joined the client streams

    # -- held-out evaluation (absent from the reference: TrainerProtocol.validate
    #    is declared in nanofed/core/interfaces.py:33 and never implemented) ----
    @torch.no_grad()
    def evaluate(self, flat: torch.Tensor, batch: int = 1000) -> dict[str, float]:
        from nanofed_amd.runtime.arena import ParameterArena

        if self._eval_inputs is None:
            return {}
        arena = ParameterArena(self.pool.layout, data=flat)
        module = arena.bind(self._eval_module)
        arena.push_int_buffers(module)
        module.eval()
        xs = self._eval_inputs.to(self.device)
        ys = self._eval_labels.to(self.device) if self._eval_labels is not None else None
        loss_sum = torch.zeros((), device=self.device)
        correct = torch.zeros((), device=self.device)
        n = 0
        for lo in range(0, len(xs), batch):
            if self.shards.kind == "image":
                x = xs[lo : lo + batch].float().div_(255.0)
                x = x.sub_(self.shards.mean).div_(self.shards.std)
                y = ys[lo : lo + batch]
            else:
                x, y = xs[lo : lo + batch, :-1], xs[lo : lo + batch, 1:]
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self._amp):
                out = module(x)
            m = len(x)
            loss_sum += classification_loss(out, y) * m
            correct += count_correct(out, y) * m
            n += m
        return {
            "loss": float(loss_sum) / max(n, 1),
            "accuracy": float(correct) / max(n, 1),
            "samples": float(n),
        }
