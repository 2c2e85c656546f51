"""Route of the reference-style training call to the hand-written kernels.

A client written against the reference (examples/mnist/run_experiment.py:21-86
there) builds a model, a ``DataLoader`` and an optimizer and calls
``TorchTrainer.train_epoch(model, dataloader, optimizer, epoch)``
(nanofed/trainer/base.py:115-198).  When that call describes exactly what the
native MNIST step implements - an ``MNISTModel`` on a HIP device, batches of 64
drawn at random from an MNIST-shaped uint8 set, plain SGD, no per-batch callbacks -
(any batch size from 1 to 1024 in fact: the kernels work on 64 rows at a time and
accumulate larger batches) the epoch runs on ``nanofed_amd.engine.mnist_native``
instead of PyTorch ops: the
model's weights are copied into a one-client arena, the epoch is one hipGraph
replay, and the trained weights are copied back into the module.  Anything else
takes the PyTorch path unchanged.

The choice belongs to the trainer's ``native`` argument (``TorchTrainer(config,
native="auto" | True | False)``); ``NANOFED_NATIVE_TRAINER=0`` turns "auto" off.
"""

from __future__ import annotations

import os
import threading
from typing import Any

import numpy as np
import torch
from torch.utils.data import DataLoader, RandomSampler, Subset

from nanofed_amd.utils import Logger

from .base import TrainingMetrics

MAX_NATIVE_BATCH = 1024  # 16 accumulated passes of 64 rows
# Client coroutines often run ``train_epoch`` in worker threads
# (``asyncio.to_thread``).  Native sessions take turns: a hipGraph capture must not
# see another thread's allocations, and the kernels of one epoch already fill the
# chip's launch queue.
_NATIVE_LOCK = threading.Lock()


def _raw_dataset(dataset: Any) -> tuple[Any, np.ndarray] | None:
    """``(dataset with .images_u8 / .labels_i64, row indices)`` or None."""
    if isinstance(dataset, Subset):
        inner = _raw_dataset(dataset.dataset)
        if inner is None:
            return None
        base, rows = inner
        return base, rows[np.asarray(dataset.indices, dtype=np.int64)]
    if hasattr(dataset, "images_u8") and hasattr(dataset, "labels_i64"):
        return dataset, np.arange(len(dataset), dtype=np.int64)
    return None


def native_epoch_unsupported_reason(trainer: Any, model: Any, dataloader: Any,
                                    optimizer: Any) -> str | None:
    """None when ``train_epoch(model, dataloader, optimizer, ...)`` is the step the native
    MNIST kernels implement, otherwise what stands in the way."""
    from nanofed_amd.models import MNISTModel

    cfg = trainer._config
    if torch.device(cfg.device).type != "cuda" or not torch.cuda.is_available():
        return f"the trainer's device is {cfg.device!r}, not a HIP device"
    if type(model) is not MNISTModel:
        return f"the model is a {type(model).__name__}, not the MNISTModel the kernels implement"
    if cfg.max_batches is not None:
        return "max_batches cuts epochs short"
    if trainer._callbacks:
        return "callbacks want per-batch host values"
    if not isinstance(dataloader, DataLoader) or dataloader.batch_size is None:
        return "the data comes from something else than a batched DataLoader"
    if not 1 <= dataloader.batch_size <= MAX_NATIVE_BATCH:
        return f"the batch size {dataloader.batch_size} is outside 1..{MAX_NATIVE_BATCH}"
    if dataloader.drop_last or not isinstance(dataloader.sampler, RandomSampler):
        return "the loader does not draw random batches with the partial last batch kept"
    raw = _raw_dataset(dataloader.dataset)
    if raw is None or tuple(raw[0].images_u8.shape[1:]) != (1, 28, 28):
        return "the dataset did not keep its uint8 images (images_u8 / labels_i64, 1x28x28)"
    if type(optimizer) is not torch.optim.SGD or len(optimizer.param_groups) != 1:
        return "the optimizer is not plain SGD with one parameter group"
    group = optimizer.param_groups[0]
    if group.get("momentum", 0) or group.get("nesterov", False) or group.get("dampening", 0):
        return "the optimizer uses momentum"
    if group.get("maximize", False):
        return "the optimizer maximizes"
    own = {id(p) for p in model.parameters()}
    if {id(p) for p in group["params"]} != own:
        return "the optimizer does not hold exactly the model's parameters"
    return None


def native_epoch_supported(trainer: Any, model: Any, dataloader: Any, optimizer: Any) -> bool:
    """True when ``train_epoch(model, dataloader, optimizer, ...)`` is the step the
    native MNIST kernels implement (and the route is not switched off)."""
    if os.environ.get("NANOFED_NATIVE_TRAINER", "1") == "0":
        return False
    return native_epoch_unsupported_reason(trainer, model, dataloader, optimizer) is None


class NativeEpochSession:
    """One client's native training state behind ``train_epoch``: a one-row arena
    pool, the device-resident shard and the kernel trainer with its epoch graph."""

    def __init__(self, dataloader: DataLoader, device: torch.device, lr: float, wd: float,
                 seed: int = 0, dp: Any = None) -> None:
        from nanofed_amd.engine.config import EngineConfig
        from nanofed_amd.engine.data import DeviceShards, FederatedData
        from nanofed_amd.engine.mnist_native import MnistCnnNativeTrainer, make_layout
        from nanofed_amd.models import MNISTModel
        from nanofed_amd.runtime.arena import ArenaPool

        base, rows = _raw_dataset(dataloader.dataset)
        images = base.images_u8[torch.from_numpy(rows)].contiguous()
        labels = base.labels_i64[torch.from_numpy(rows)].contiguous()
        n = len(rows)
        self.samples = n
        self.lr, self.wd = float(lr), float(wd)
        self.device = device
        batch = int(dataloader.batch_size)
        cfg = EngineConfig(
            model="mnist_cnn", num_clients=1, samples_per_client=n, batch_size=batch,
            local_epochs=1, learning_rate=self.lr, weight_decay=self.wd, dtype="bf16",
            backend="native", seed=seed, eval_samples=0, dp=dp,
        )
        data = FederatedData("image", images, labels, [np.arange(n)], None, None,
                             (float(base.mean),), (float(base.std),))
        self.layout = make_layout(MNISTModel())
        self.pool = ArenaPool(self.layout, 1, device)
        self.shards = DeviceShards(data, [0], device, batch, seed)
        self.trainer = MnistCnnNativeTrainer(cfg, self.pool, self.shards, data)
        self.steps = self.shards.steps_per_epoch(0)
        self.epochs_run = 0

    @torch.no_grad()
    def train_epoch(self, model: torch.nn.Module, epoch: int) -> TrainingMetrics:
        from nanofed_amd.engine.trainers import M_ACC, M_LOSS

        with _NATIVE_LOCK:
            arena = self.pool[0]
            arena.load_state_dict(model.state_dict())
            metrics = self.trainer.train_round(self.epochs_run)  # one local epoch
            self.epochs_run += 1
            model.load_state_dict(arena.state_dict(self.device))
            loss, acc = metrics[0, M_LOSS].item(), metrics[0, M_ACC].item()
        return TrainingMetrics(loss=float(loss), accuracy=float(acc), epoch=epoch,
                               batch=max(self.steps - 1, 0), samples_processed=self.samples,
                               backend="native")


def _loader_signature(dataloader: DataLoader) -> tuple:
    """What a session was built from: the dataset object, its length and the batch size.
    A session is only reused for the loader it was built for while these still hold."""
    dataset = dataloader.dataset
    return (id(dataset), len(dataset), int(dataloader.batch_size))


def native_session_for(trainer: Any, dataloader: DataLoader, optimizer: Any,
                       factory: Any = None) -> NativeEpochSession:
    """The session of ``dataloader`` (built once: it places the shard in HBM and
    captures the epoch graph); a new learning rate or weight decay builds a new one,
    because they are arguments of the captured kernels.

    Sessions are kept in a ``WeakKeyDictionary`` on the loader OBJECT: a session lives
    exactly as long as its loader (a client that builds a new loader every round gets a
    new session and the old one - shard, arena, graphs - is released with the old
    loader), and an address that the allocator hands out again cannot be mistaken for
    the loader that used to live there."""
    import weakref

    group = optimizer.param_groups[0]
    lr, wd = float(group["lr"]), float(group.get("weight_decay", 0.0))
    cache = trainer.__dict__.get("_native_sessions")
    if cache is None:
        cache = trainer.__dict__["_native_sessions"] = weakref.WeakKeyDictionary()
    signature = _loader_signature(dataloader)
    session = cache.get(dataloader)
    if (session is None or session.lr != lr or session.wd != wd
            or getattr(session, "signature", signature) != signature):
        Logger().info(
            f"train_epoch runs on the native MNIST kernels (bf16 compute, fp32 master weights; "
            f"{len(dataloader.dataset)} samples, lr {lr}, one hipGraph per epoch)"
        )
        make = factory or NativeEpochSession
        with _NATIVE_LOCK:
            session = make(dataloader, torch.device(trainer._config.device), lr, wd)
        session.signature = signature
        cache[dataloader] = session
    return session


# ---------------------------------------------------------------------------------------
# DP-SGD: PrivateTrainer.train_batch on the native per-sample kernels


def _plain_sgd_reason(model: Any, optimizer: Any) -> str | None:
    if type(optimizer) is not torch.optim.SGD or len(optimizer.param_groups) != 1:
        return "the optimizer is not plain SGD with one parameter group"
    group = optimizer.param_groups[0]
    if group.get("momentum", 0) or group.get("nesterov", False) or group.get("dampening", 0):
        return "the optimizer uses momentum"
    if group.get("maximize", False):
        return "the optimizer maximizes"
    own = {id(p) for p in model.parameters()}
    if {id(p) for p in group["params"]} != own:
        return "the optimizer does not hold exactly the model's parameters"
    return None


def native_dp_batch_unsupported_reason(trainer: Any, model: Any, inputs: Any, targets: Any,
                                       optimizer: Any) -> str | None:
    """None when ``PrivateTrainer.train_batch(model, (inputs, targets), optimizer)`` is the
    step the native DP-SGD kernels implement (per-sample clipping, Gaussian noise drawn on
    the device), otherwise what stands in the way."""
    from nanofed_amd.models import MNISTModel
    from nanofed_amd.privacy.config import NoiseType

    cfg = trainer._config
    if torch.device(cfg.device).type != "cuda" or not torch.cuda.is_available():
        return f"the trainer's device is {cfg.device!r}, not a HIP device"
    if trainer._clip_mode != "per_sample":
        return "clip_mode is 'batch' (the kernels clip every example's gradient)"
    if type(model) is not MNISTModel:
        return f"the model is a {type(model).__name__}, not the MNISTModel the kernels implement"
    if trainer._privacy_config.noise_type != NoiseType.GAUSSIAN:
        return "the noise is not Gaussian"
    if not trainer._device_noise:
        return "a host noise generator was passed (its stream is drawn with torch.randn)"
    if not torch.is_tensor(inputs) or inputs.dim() < 2 or inputs[0].numel() != 784:
        return "the batch is not a tensor of 28x28 images"
    if not 1 <= inputs.shape[0] <= 64 or inputs.shape[0] != targets.shape[0]:
        return f"the batch has {inputs.shape[0]} rows (1..64 are covered)"
    return _plain_sgd_reason(model, optimizer)


class NativeDPStep:
    """One model's native DP-SGD state behind ``PrivateTrainer.train_batch``: a one-row
    arena, the kernel buffers, and a staging area for the batch (normalised float images,
    labels).  A step is one call of the fused DP sequence of ``csrc/hip/mnist_cnn.hip``:
    forward, a backward pass that measures every example's gradient norm, clip factors,
    the re-weighted pass, Philox Gaussian noise of std sigma * C on the sum, SGD.

    The model's weights are copied into the arena when they changed since the last step
    (somebody else wrote them) and copied back after every step."""

    def __init__(self, device: torch.device, clip: float, sigma: float, seed: int) -> None:
        from nanofed_amd.engine.config import DPConfig, EngineConfig
        from nanofed_amd.engine.mnist_native import BATCH, NativeBuffers, make_layout
        from nanofed_amd.models import MNISTModel
        from nanofed_amd.runtime.arena import ArenaPool

        self.device = device
        self.layout = make_layout(MNISTModel())
        self.pool = ArenaPool(self.layout, 1, device)
        cfg = EngineConfig(model="mnist_cnn", num_clients=1, batch_size=BATCH, learning_rate=0.0,
                           backend="native", eval_samples=0)
        dummy = torch.zeros(1, 784, dtype=torch.uint8, device=device)
        self.labels = torch.zeros(BATCH, dtype=torch.int32, device=device)
        self.buf = NativeBuffers(
            self.layout, 1, device, dummy, self.labels, self.pool.data, self.layout.numel,
            self.pool.grad, 1, 0.0, 1.0, cfg, train=True, seed=seed,
            dp=DPConfig(max_gradient_norm=clip, noise_multiplier=sigma, mode="per_sample"),
        )
        self.buf.labels = self.labels
        self.buf.ctx.labels = self.labels.data_ptr()
        self.images = torch.zeros(BATCH, 784, dtype=torch.float32, device=device)
        self.buf.ctx.images_f32 = self.images.data_ptr()
        self.buf.idx.copy_(torch.arange(BATCH, dtype=torch.int32).view(1, BATCH))
        self._rows = -1
        self._synced: tuple | None = None
        self._pairs: list[tuple[torch.Tensor, torch.Tensor]] | None = None
        self._pairs_for: tuple | None = None
        self.steps = 0

    def _versions(self, model: torch.nn.Module) -> tuple:
        return tuple((id(p), p._version) for p in model.parameters())

    def _param_pairs(self, model: torch.nn.Module) -> list[tuple[torch.Tensor, torch.Tensor]]:
        """(module parameter, view of the arena row in the parameter's own index order) for
        every parameter: the two tensors stored in the kernels' order (conv2.weight as
        [oc][kh][kw][c], fc1.weight with NHWC-flattened columns) are permuted VIEWS, so a
        sync in either direction is eight strided copies - no index gathers, no
        ``load_state_dict`` machinery per step."""
        key = tuple(id(p) for p in model.parameters())
        if self._pairs is None or self._pairs_for != key:
            row = self.pool.data[0]
            pairs = []
            for name, p in model.named_parameters():
                seg = self.layout[name]
                flat = row[seg.offset : seg.offset + seg.numel]
                if name == "conv2.weight":
                    view = flat.view(64, 3, 3, 32).permute(0, 3, 1, 2)
                elif name == "fc1.weight":
                    view = flat.view(128, 144, 64).permute(0, 2, 1)
                    p = p.view(128, 64, 144)
                else:
                    view = flat.view(p.shape)
                pairs.append((p, view))
            self._pairs, self._pairs_for = pairs, key
        return self._pairs

    @torch.no_grad()
    def step(self, model: torch.nn.Module, inputs: torch.Tensor, targets: torch.Tensor,
             lr: float, wd: float) -> tuple[float, float]:
        """One DP-SGD step; returns (mean loss, accuracy) of the batch."""
        n = inputs.shape[0]
        with _NATIVE_LOCK:
            pairs = self._param_pairs(model)
            if self._synced != self._versions(model):
                for p, view in pairs:
                    view.copy_(p)
                self.buf.pack()
            self.images[:n].copy_(inputs.reshape(n, 784))
            self.labels[:n].copy_(targets.reshape(n))
            if n != self._rows:
                self.buf.nvalid.fill_(n)
                self.buf.ntotal.fill_(n)
                self._rows = n
            ctx = self.buf.ctx
            ctx.lr, ctx.wd = float(lr), float(wd)
            ctx.train = 1 if model.training else 0
            self.buf.metrics.zero_()
            self.buf.train_steps(0, 1)
            self.steps += 1
            for p, view in pairs:
                p.copy_(view)
            self._synced = self._versions(model)
            loss, acc = self.buf.metrics[0, :2].tolist()
        return float(loss), float(acc)
