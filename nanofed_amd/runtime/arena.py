"""Flat parameter arenas.

A model's whole ``state_dict`` lives in ONE contiguous fp32 device buffer (the
arena); a layout table maps each state-dict key to an offset and a shape.
Everything the federation does to a model between training steps then becomes
one launch / one collective over one buffer, independent of how many tensors
the model has:

* global-model broadcast (reference: GET /model, nanofed/communication/http/server.py:154-208)
  is one copy or one RCCL broadcast;
* FedAvg (nanofed/server/aggregator/fedavg.py:56-63) is one fused weighted sum
  and one all-reduce;
* SGD (nanofed/trainer/base.py:142,150) is one multi-tensor kernel;
* clipping / noise / validation are one or two reduction kernels;
* a checkpoint (nanofed/server/model_manager/manager.py:99-142) is one D2H copy.

Segments are aligned to 64 floats (256 B) so every tensor view can be read
with 16-byte vector loads.  Trainable parameters come first (the prefix
``[0, trainable_numel)`` has a gradient arena of the same length), buffers
(BatchNorm statistics and the like) follow; integer buffers are stored as
fp32 values and cast back on export, which is also how the reference averages
them (fedavg.py:39-44).
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Iterator, Mapping

import torch
import torch.nn as nn

from nanofed_amd.core.exceptions import DeviceError

ALIGN = 64  # floats


def _round_up(n: int, a: int = ALIGN) -> int:
    return (n + a - 1) // a * a


@dataclass(frozen=True)
class Segment:
    name: str
    offset: int
    numel: int
    shape: tuple[int, ...]
    dtype: torch.dtype
    trainable: bool


class ArenaLayout:
    """Immutable key -> segment table shared by every arena of one model."""

    def __init__(
        self, segments: list[Segment], perms: dict[str, torch.Tensor] | None = None,
        view_strides: dict[str, tuple[int, ...]] | None = None,
    ) -> None:
        self.segments = tuple(segments)
        # storage permutations: stored_flat = canonical_flat[perm] for the
        # tensors a kernel family wants in its own element order
        self.perms = dict(perms or {})
        # permutations that are plain dimension reorderings can still be viewed as
        # ordinary (strided) tensors of the canonical shape: name -> strides
        self.view_strides = dict(view_strides or {})
        self.inv_perms = {k: torch.argsort(v) for k, v in self.perms.items()}
        self._by_name = {s.name: s for s in segments}
        if len(self._by_name) != len(segments):
            raise DeviceError("duplicate key in arena layout")
        train = [s for s in segments if s.trainable]
        self.trainable_numel = (
            _round_up(train[-1].offset + train[-1].numel) if train else 0
        )
        last = segments[-1] if segments else None
        self.numel = _round_up(last.offset + last.numel) if last else 0
        self.payload_numel = sum(s.numel for s in segments)

    @classmethod
    def from_module(cls, module: nn.Module) -> "ArenaLayout":
        sd = module.state_dict()
        param_names = {n for n, _ in module.named_parameters()}
        ordered = [k for k in sd if k in param_names] + [
            k for k in sd if k not in param_names
        ]
        segs, off = [], 0
        for key in ordered:
            t = sd[key]
            segs.append(
                Segment(key, off, t.numel(), tuple(t.shape), t.dtype, key in param_names)
            )
            off = _round_up(off + t.numel())
        return cls(segs)

    @classmethod
    def from_state_dict(cls, sd: Mapping[str, torch.Tensor]) -> "ArenaLayout":
        segs, off = [], 0
        for key, t in sd.items():
            t = torch.as_tensor(t)
            segs.append(Segment(key, off, t.numel(), tuple(t.shape), t.dtype, True))
            off = _round_up(off + t.numel())
        return cls(segs)

    def __getitem__(self, name: str) -> Segment:
        return self._by_name[name]

    def __contains__(self, name: str) -> bool:
        return name in self._by_name

    def keys(self) -> list[str]:
        return [s.name for s in self.segments]

    def signature(self) -> tuple:
        return tuple((s.name, s.offset, s.shape) for s in self.segments)

    def with_perms(self, perms: dict[str, torch.Tensor]) -> "ArenaLayout":
        """Same table, with some tensors stored in a permuted element order."""
        for name, perm in perms.items():
            if perm.numel() != self[name].numel:
                raise DeviceError(f"permutation of {name} has the wrong length")
        return ArenaLayout(list(self.segments), perms)

    def with_channels_last(self) -> "ArenaLayout":
        """Store every trainable 4-D tensor (convolution weights ``[O, I, H, W]``)
        in ``[O, H, W, I]`` order.  Such a tensor is viewed with channels-last
        strides, so a bound module sees the usual shape while the convolution
        library gets NHWC weights without a transposition per call."""
        perms, strides = dict(self.perms), dict(self.view_strides)
        for s in self.segments:
            if s.trainable and len(s.shape) == 4 and s.name not in perms:
                o, i, h, w = s.shape
                idx = torch.arange(s.numel).view(o, i, h, w)
                perms[s.name] = idx.permute(0, 2, 3, 1).reshape(-1)
                strides[s.name] = (h * w * i, 1, w * i, i)
        return ArenaLayout(list(self.segments), perms, strides)

    def bindable(self) -> bool:
        """True when every permuted tensor still has a strided view."""
        return all(name in self.view_strides for name in self.perms)

    def view_of(self, flat: torch.Tensor, seg: Segment) -> torch.Tensor:
        """The tensor of ``seg`` inside ``flat``: canonical shape, and canonical
        VALUES whenever the storage order is expressible as strides."""
        strides = self.view_strides.get(seg.name)
        if strides is not None:
            return torch.as_strided(flat, seg.shape, strides, flat.storage_offset() + seg.offset)
        return flat[seg.offset : seg.offset + seg.numel].view(seg.shape)


class ArenaStateDict(dict):
    """``state_dict``-shaped dictionary whose values are views into one flat
    buffer.  Aggregators and validators recognise it (``flat`` / ``layout``)
    and run single-kernel paths over the buffer."""

    def __init__(self, layout: ArenaLayout, flat: torch.Tensor) -> None:
        if flat.dim() != 1 or flat.numel() != layout.numel:
            raise DeviceError("flat buffer does not match the arena layout")
        super().__init__((s.name, layout.view_of(flat, s)) for s in layout.segments)
        self.layout = layout
        self.flat = flat

    def with_flat(self, flat: torch.Tensor) -> "ArenaStateDict":
        return ArenaStateDict(self.layout, flat)

    def per_tensor_norms(self) -> torch.Tensor:
        """L2 norm of every tensor, as one device vector."""
        views = [v.reshape(-1) for v in self.values()]
        return torch.stack(list(torch._foreach_norm(views))) if views else torch.zeros(0)

    def export(self, device: str | torch.device = "cpu") -> dict[str, torch.Tensor]:
        """Detached copy in the segments' original dtypes (checkpoint format)."""
        host = self.flat.detach().to(device)
        out = {}
        for s in self.layout.segments:
            seg = host[s.offset : s.offset + s.numel]
            inv = self.layout.inv_perms.get(s.name)
            if inv is not None:
                seg = torch.take(seg, inv.to(seg.device))  # a new tensor already
                out[s.name] = seg.view(s.shape) if s.dtype == seg.dtype else seg.view(s.shape).to(s.dtype)
            else:
                out[s.name] = seg.view(s.shape).to(s.dtype, copy=True)
        return out


class ParameterArena:
    """One model's parameters (and buffers) in a flat fp32 buffer, with an
    optional gradient buffer over the trainable prefix."""

    def __init__(
        self,
        layout: ArenaLayout,
        device: str | torch.device = "cpu",
        data: torch.Tensor | None = None,
        grad: torch.Tensor | None = None,
        with_grad: bool = False,
    ) -> None:
        self.layout = layout
        dev = torch.device(device) if data is None else data.device
        if data is None:
            data = torch.zeros(layout.numel, dtype=torch.float32, device=dev)
        if data.dtype != torch.float32 or data.numel() != layout.numel or data.dim() != 1:
            raise DeviceError("arena storage must be a flat fp32 buffer of layout.numel")
        self.data = data
        if grad is None and with_grad:
            grad = torch.zeros(layout.trainable_numel, dtype=torch.float32, device=dev)
        self.grad = grad

    # -- construction -----------------------------------------------------
    @classmethod
    def from_module(
        cls, module: nn.Module, device: str | torch.device | None = None,
        with_grad: bool = False,
    ) -> "ParameterArena":
        layout = ArenaLayout.from_module(module)
        if device is None:
            first = next(iter(module.state_dict().values()), None)
            device = first.device if first is not None else "cpu"
        arena = cls(layout, device, with_grad=with_grad)
        arena.load_state_dict(module.state_dict())
        return arena

    @property
    def device(self) -> torch.device:
        return self.data.device

    @property
    def trainable(self) -> torch.Tensor:
        return self.data[: self.layout.trainable_numel]

    # -- state-dict boundary ----------------------------------------------
    def views(self) -> ArenaStateDict:
        return ArenaStateDict(self.layout, self.data)

    def grad_views(self) -> dict[str, torch.Tensor]:
        if self.grad is None:
            raise DeviceError("arena has no gradient buffer")
        return {
            s.name: self.layout.view_of(self.grad, s)
            for s in self.layout.segments
            if s.trainable
        }

    def load_state_dict(self, state: Mapping[str, object]) -> None:
        """Copy a ``state_dict`` (tensors or nested lists) into the arena."""
        missing = [k for k in self.layout.keys() if k not in state]
        if missing:
            raise DeviceError(f"state is missing keys: {missing}")
        if isinstance(state, ArenaStateDict):
            if state.layout is self.layout:
                self.data.copy_(state.flat)
                return
            state = state.export(state.flat.device)  # canonical element order
        # stage on the host and move once: one H2D copy instead of one per key
        same_dev = all(
            isinstance(v, torch.Tensor) and v.device == self.device
            for v in state.values()
        )
        stage = self.data if same_dev else torch.zeros(
            self.layout.numel, dtype=torch.float32
        )
        for s in self.layout.segments:
            src = torch.as_tensor(state[s.name])
            if src.numel() != s.numel:
                raise DeviceError(
                    f"{s.name}: expected {s.shape}, got {tuple(src.shape)}"
                )
            flat_src = src.detach().reshape(-1).to(torch.float32)
            perm = self.layout.perms.get(s.name)
            if perm is not None:
                flat_src = flat_src[perm.to(flat_src.device)]
            stage[s.offset : s.offset + s.numel].copy_(flat_src)
        if stage is not self.data:
            self.data.copy_(stage)

    def state_dict(self, device: str | torch.device = "cpu") -> dict[str, torch.Tensor]:
        return self.views().export(device)

    # -- module binding ---------------------------------------------------
    def bind(self, module: nn.Module) -> nn.Module:
        """Re-point ``module``'s parameters, gradients and float buffers at
        the arena, so autograd and the arena kernels operate on the same
        memory.  Integer buffers keep their own storage (``sync_buffers``)."""
        if not self.layout.bindable():
            raise DeviceError(
                "an arena with permuted storage cannot back a torch module; "
                "use state_dict()/load_state_dict() at the boundary"
            )
        views = self.views()
        gviews = self.grad_views() if self.grad is not None else {}
        for name, p in module.named_parameters():
            p.data = views[name]
            if name in gviews:
                p.grad = gviews[name]
        for name, b in module.named_buffers():
            if name in views and b.dtype.is_floating_point:
                _set_buffer(module, name, views[name])
        return module

    def pull_int_buffers(self, module: nn.Module, skip: frozenset[str] = frozenset()) -> None:
        """module integer buffers -> arena (before aggregation); ``skip``: names that
        are maintained in the arena itself."""
        views = self.views()
        for name, b in module.named_buffers():
            if name in views and name not in skip and not b.dtype.is_floating_point:
                views[name].copy_(b.to(torch.float32))

    def push_int_buffers(self, module: nn.Module, skip: frozenset[str] = frozenset()) -> None:
        """arena -> module integer buffers (after a new global model)."""
        views = self.views()
        for name, b in module.named_buffers():
            if name in views and name not in skip and not b.dtype.is_floating_point:
                b.copy_(views[name].round().to(b.dtype))


def _set_buffer(module: nn.Module, dotted: str, value: torch.Tensor) -> None:
    owner = module
    *path, leaf = dotted.split(".")
    for part in path:
        owner = getattr(owner, part)
    owner._buffers[leaf] = value


class ArenaPool:
    """``count`` arenas of one layout in one ``[count, numel]`` allocation -
    the simulated clients that share a GPU.  Row k is client k's model, so a
    broadcast of the global model is one expand-copy and the local part of
    FedAvg is one weighted sum over rows."""

    def __init__(
        self, layout: ArenaLayout, count: int, device: str | torch.device,
        with_grad: bool = True,
    ) -> None:
        if count < 1:
            raise DeviceError("an arena pool needs at least one arena")
        self.layout = layout
        self.count = count
        self.data = torch.zeros(count, layout.numel, dtype=torch.float32, device=device)
        self.grad = (
            torch.zeros(count, layout.trainable_numel, dtype=torch.float32, device=device)
            if with_grad
            else None
        )
        self._arenas = [
            ParameterArena(
                layout, data=self.data[k],
                grad=self.grad[k] if self.grad is not None else None,
            )
            for k in range(count)
        ]

    def __len__(self) -> int:
        return self.count

    def __getitem__(self, k: int) -> ParameterArena:
        return self._arenas[k]

    def __iter__(self) -> Iterator[ParameterArena]:
        return iter(self._arenas)

    def broadcast_from(self, flat: torch.Tensor) -> None:
        """Every client starts the round from the global model."""
        if flat.numel() != self.layout.numel:
            raise DeviceError("global buffer does not match the pool layout")
        self.data.copy_(flat.unsqueeze(0).expand(self.count, -1))

    def rows(self) -> list[torch.Tensor]:
        return [a.data for a in self._arenas]
