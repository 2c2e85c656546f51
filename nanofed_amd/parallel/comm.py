"""Collective transport of the device plane: one process per GPU,
``torch.distributed`` with the ``nccl`` backend (RCCL over xGMI on ROCm) or
``gloo`` (CPU tests, multi-process rehearsal).

The reference moves models as JSON over HTTP (GET /model, POST /update:
nanofed/communication/http/server.py:154-298, client.py:103-211).  Here the
same logical traffic is three collectives over flat arenas:

* ``broadcast``         global model, server -> all clients
* ``allreduce_sum``     sum of pre-scaled client arenas (FedAvg reduce)
* ``allgather``         a few floats per client (sample counts, loss, accuracy,
                        participation flag, privacy spent)

xGMI is point-to-point (7 links per GPU), so a ring all-reduce of a large
arena is bound by one link.  ``allreduce_sum`` therefore switches by size:
small arenas (latency-bound) use a single ``all_reduce``; large arenas use
``reduce_scatter`` + ``all_gather`` on a padded buffer, which lets RCCL drive
all peers at once and lets the caller fuse work on the 1/world shard between
the two phases (``shard_hook``).

``gloo`` only ever sees CPU tensors: device tensors are staged through host
memory with synchronous copies.
"""

from __future__ import annotations

import datetime
import os
from typing import Callable

import torch
import torch.distributed as dist

from nanofed_amd.core.exceptions import DeviceError

# arenas at least this large take the reduce-scatter + all-gather path
RS_AG_THRESHOLD_BYTES = 32 * 1024 * 1024


class Communicator:
    """Rank/world bookkeeping plus the collectives used by the engine.

    A world of one needs no process group; every collective is then the
    identity, so single-GPU runs have no distributed dependency.
    """

    def __init__(
        self,
        rank: int = 0,
        world_size: int = 1,
        local_rank: int = 0,
        backend: str | None = None,
        device: torch.device | None = None,
        group: dist.ProcessGroup | None = None,
        owns_group: bool = False,
        force_collectives: bool = False,
    ) -> None:
        self.rank = rank
        self.world_size = world_size
        self.local_rank = local_rank
        self.backend = backend
        self.device = device or torch.device("cpu")
        self.group = group
        self._owns_group = owns_group
        # a world of one skips every collective unless this is set
        self._single = world_size == 1 and not (force_collectives and group is not None)
        self._pad: dict[int, torch.Tensor] = {}

    # -- construction -----------------------------------------------------
    @classmethod
    def from_env(
        cls,
        backend: str | None = None,
        device: str | torch.device | None = None,
        timeout_s: float = 600.0,
    ) -> "Communicator":
        """Join the job described by RANK / WORLD_SIZE / LOCAL_RANK /
        MASTER_ADDR / MASTER_PORT (the ``torch.distributed.run`` contract)."""
        # dmabuf IPC for RCCL's peer-memory exchange (read when the runtime starts, so
        # it only helps when nothing has touched the GPU yet; launchers should export it)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
        if device is None:
            n_dev = torch.cuda.device_count()
            if n_dev > 0:
                device = torch.device("cuda", local_rank % n_dev)
            else:
                device = torch.device("cpu")
        device = torch.device(device)
        if device.type == "cuda":
            torch.cuda.set_device(device)
        if backend is None:
            backend = "nccl" if device.type == "cuda" else "gloo"
        # NANOFED_FORCE_GROUP=1 builds the process group even for a single rank, so
        # the collectives really go through RCCL / gloo (a one-GPU rehearsal of the
        # multi-GPU code path)
        force = os.environ.get("NANOFED_FORCE_GROUP", "0") == "1"
        if world == 1 and not force:
            return cls(0, 1, local_rank, None, device)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        owns = False
        if not dist.is_initialized():
            kwargs = {}
            if backend == "nccl":
                kwargs["device_id"] = device
            dist.init_process_group(
                backend=backend, rank=rank, world_size=world,
                timeout=datetime.timedelta(seconds=timeout_s), **kwargs,
            )
            owns = True
        return cls(rank, world, local_rank, backend, device, dist.group.WORLD, owns, force)

    def close(self) -> None:
        """Leave the job: the ranks meet once more (so that none tears its connections down
        while another still uses them), then the group is destroyed."""
        if self._owns_group and dist.is_initialized():
            try:
                self.barrier()
            except Exception:  # noqa: BLE001 - a peer that is gone already cannot be waited for
                pass
            dist.destroy_process_group()
            self._owns_group = False

    def finish(self, code: int = 0) -> None:
        """End a multi-process job's process: ``close()``, flush the standard streams and
        leave without running the interpreter's and the libraries' exit handlers.  The
        collective libraries keep helper threads of their own, and their tear-down at
        interpreter exit has been seen to abort a rank that had finished all of its work
        (``terminate called without an active exception``, about one three-rank gloo job
        in fifteen); a launcher reads that as a failed job.  With one process this is a
        plain return - there is nothing to race with."""
        multi = self.world_size > 1
        self.close()
        if multi:
            import sys

            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(code)

    # -- helpers ----------------------------------------------------------
    @property
    def is_root(self) -> bool:
        return self.rank == 0

    def _host_staged(self, t: torch.Tensor) -> bool:
        return self.backend == "gloo" and t.device.type != "cpu"

    def _run(self, t: torch.Tensor, fn: Callable[[torch.Tensor], None]) -> None:
        if self._host_staged(t):
            host = t.detach().to("cpu")  # synchronous D2H
            fn(host)
            t.copy_(host)  # synchronous H2D
        else:
            fn(t)

    # -- collectives ------------------------------------------------------
    def barrier(self) -> None:
        if self._single:
            return
        if self.backend == "nccl":
            dist.barrier(group=self.group, device_ids=[self.device.index or 0])
        else:
            dist.barrier(group=self.group)

    def broadcast(self, flat: torch.Tensor, src: int = 0) -> torch.Tensor:
        if not self._single:
            self._run(flat, lambda t: dist.broadcast(t, src=src, group=self.group))
        return flat

    def allreduce_sum(
        self,
        flat: torch.Tensor,
        algo: str = "auto",
        shard_hook: Callable[[torch.Tensor, int], None] | None = None,
    ) -> torch.Tensor:
        """In-place sum over ranks.  ``algo``: ``auto`` | ``allreduce`` |
        ``rs_ag``.  ``shard_hook(shard, start)`` runs on this rank's reduced
        1/world slice between reduce-scatter and all-gather (``rs_ag`` only)."""
        if algo not in ("auto", "allreduce", "rs_ag"):
            raise DeviceError(f"unknown all-reduce algorithm {algo!r}")
        if self._single:
            if shard_hook is not None:
                shard_hook(flat, 0)
            return flat
        if algo == "auto":
            big = flat.numel() * flat.element_size() >= RS_AG_THRESHOLD_BYTES
            algo = "rs_ag" if (big or shard_hook is not None) else "allreduce"
        if algo == "allreduce":
            self._run(
                flat, lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            )
            return flat
        self._run(flat, lambda t: self._rs_ag(t, shard_hook))
        return flat

    def _rs_ag(
        self, t: torch.Tensor, shard_hook: Callable[[torch.Tensor, int], None] | None
    ) -> None:
        w = self.world_size
        n = t.numel()
        per = (n + w - 1) // w
        per = (per + 63) // 64 * 64
        padded = per * w
        if padded == n:
            buf = t
        else:
            buf = self._pad.get(padded)
            if buf is None or buf.device != t.device or buf.dtype != t.dtype:
                buf = torch.zeros(padded, dtype=t.dtype, device=t.device)
                self._pad[padded] = buf
            buf[:n].copy_(t)
            buf[n:].zero_()
        shard = torch.empty(per, dtype=t.dtype, device=t.device)
        dist.reduce_scatter_tensor(shard, buf, op=dist.ReduceOp.SUM, group=self.group)
        if shard_hook is not None:
            shard_hook(shard, self.rank * per)
        dist.all_gather_into_tensor(buf, shard, group=self.group)
        if buf is not t:
            t.copy_(buf[:n])

    def allgather(self, local: torch.Tensor) -> torch.Tensor:
        """``[world, *local.shape]`` stack of every rank's tensor."""
        if self._single:
            return local.unsqueeze(0).clone()
        staged = self._host_staged(local)
        src = local.detach().to("cpu") if staged else local.contiguous()
        flat = src.reshape(-1)
        out = torch.empty(self.world_size * flat.numel(), dtype=src.dtype, device=src.device)
        dist.all_gather_into_tensor(out, flat, group=self.group)
        out = out.view(self.world_size, *src.shape)
        return out.to(local.device) if staged else out

    def max_over_ranks(self, value: float) -> float:
        if self._single:
            return float(value)
        t = torch.tensor([float(value)], dtype=torch.float64)
        if self.backend == "nccl":
            t = t.to(self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())
