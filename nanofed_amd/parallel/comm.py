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
``reduce_scatter`` + ``all_gather``, which lets RCCL drive all peers at once.
The buffer handed in should be ``padded_numel`` long (the engine allocates it
so), then neither phase copies.

Both of these leave the order of the additions to the library, so the bits of
the result depend on the number of ranks.  ``exchange_row_shards`` +
``allgather_shards`` are the transport of the *ordered* reduce: every rank
receives slice ``rank`` of EVERY client row (one all-to-all), adds the rows in
client order with the same kernel a single process uses, and the reduced
slices are gathered - the same traffic as reduce-scatter + all-gather for one
client per rank, and a result that is bit-identical for 1, 2, ... ranks
(reference: the server adds the updates in one fixed loop,
nanofed/server/aggregator/fedavg.py:56-63).

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
        self._pad: dict[str, torch.Tensor] = {}

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
        comm = cls(rank, world, local_rank, backend, device, dist.group.WORLD, owns, force)
        if owns:
            comm._close_at_exit()
        return comm

    def _close_at_exit(self) -> None:
        """A process must not reach interpreter shutdown with a live process group: the
        group's helper threads (gloo's workers are ``std::thread`` members) are then still
        running while static objects are destroyed, and about one three-rank gloo job in
        thirteen had a rank end with ``terminate called without an active exception``
        (exit status -6) after all of its work was done - reproduced on the CPU with 3 of 40
        jobs that left without ``destroy_process_group`` against 0 of 140 that destroyed
        the group first.  ``close()`` is therefore also registered with ``atexit`` (which
        runs while the interpreter and all threads are still intact) for callers that
        forget it."""
        import atexit
        import sys
        import weakref

        ref = weakref.ref(self)

        def _guard() -> None:
            comm = ref()
            if comm is not None:
                try:
                    comm.close(failed=getattr(sys, "last_type", None) is not None)
                except Exception:  # noqa: BLE001 - nothing useful can be done at exit
                    pass

        atexit.register(_guard)

    CLOSE_WAIT_S = 10.0

    def close(self, failed: bool = False) -> None:
        """Leave the job: the ranks meet once more (so that none tears its connections
        down while another still uses them), then the group is destroyed.  The meeting
        is a count in the rendezvous store, polled for at most ``CLOSE_WAIT_S`` seconds:
        it launches nothing on the device and cannot hang on a peer that is gone
        (a collective barrier would sit there until the group's own time-out).
        ``failed=True`` - the job is ending because of an error - skips it."""
        if self._owns_group and dist.is_initialized():
            if not failed and self.world_size > 1:
                self._meet("nanofed/close")
            dist.destroy_process_group()
            self._owns_group = False

    def _meet(self, key: str) -> bool:
        """True when every rank arrived at ``key`` within ``CLOSE_WAIT_S``."""
        import time

        try:
            store = dist.distributed_c10d._get_default_store()
            store.add(key, 1)
            deadline = time.monotonic() + self.CLOSE_WAIT_S
            while time.monotonic() < deadline:
                if store.add(key, 0) >= self.world_size:
                    return True
                time.sleep(0.005)
        except Exception:  # noqa: BLE001 - the store lives on rank 0, which may be gone
            pass
        return False

    def finish(self, code: int = 0) -> None:
        """End a job's process in an orderly way; the LAST call of a worker: ``close()``
        (the group is destroyed before the interpreter shuts down - see
        ``_close_at_exit`` for what happens otherwise), then logging handlers and the
        standard streams are flushed.  With gloo or a single process this returns and the
        process ends the ordinary way, exit handlers included.

        Several RCCL ranks leave through ``os._exit`` after the flush: the tear-down of a
        multi-GPU RCCL job at interpreter exit cannot be rehearsed on a one-GPU machine,
        and a rank that has finished all of its work must not be turned into a failed job
        by it.  (No such failure has been observed; this is a precaution, not a fix.)"""
        multi = self.world_size > 1
        backend = self.backend
        self.close(failed=code != 0)
        import logging
        import sys

        logging.shutdown()
        sys.stdout.flush()
        sys.stderr.flush()
        if multi and backend == "nccl":
            os._exit(code)

    # -- helpers ----------------------------------------------------------
    @property
    def is_root(self) -> bool:
        return self.rank == 0

    @property
    def collective(self) -> bool:
        """True when the collectives really go through a process group (several ranks, or
        a one-rank group forced for a rehearsal); False when they are the identity."""
        return not self._single

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

    def shard_numel(self, numel: int) -> int:
        """Elements of one rank's slice of a ``numel``-long arena (a multiple of 64)."""
        per = (numel + self.world_size - 1) // self.world_size
        return (per + 63) // 64 * 64

    def padded_numel(self, numel: int) -> int:
        """Length to allocate for a buffer of ``numel`` elements so that the sliced
        collectives work on it in place."""
        return self.shard_numel(numel) * self.world_size

    def allreduce_sum(self, flat: torch.Tensor, algo: str = "auto") -> torch.Tensor:
        """In-place sum over ranks.  ``algo``: ``auto`` | ``allreduce`` | ``rs_ag``."""
        if algo not in ("auto", "allreduce", "rs_ag"):
            raise DeviceError(f"unknown all-reduce algorithm {algo!r}")
        if self._single:
            return flat
        if algo == "auto":
            big = flat.numel() * flat.element_size() >= RS_AG_THRESHOLD_BYTES
            algo = "rs_ag" if big else "allreduce"
        if algo == "allreduce":
            self._run(
                flat, lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            )
            return flat
        self._run(flat, self._rs_ag)
        return flat

    def _scratch(self, key: str, numel: int, like: torch.Tensor) -> torch.Tensor:
        buf = self._pad.get(key)
        if buf is None or buf.numel() != numel or buf.device != like.device or buf.dtype != like.dtype:
            buf = torch.zeros(numel, dtype=like.dtype, device=like.device)
            self._pad[key] = buf
        return buf

    def _rs_ag(self, t: torch.Tensor) -> None:
        n = t.numel()
        per = self.shard_numel(n)
        padded = per * self.world_size
        if padded == n:
            buf = t  # the usual case: the caller allocated padded_numel elements
        else:
            buf = self._scratch("rs_ag", padded, t)
            buf[:n].copy_(t)
            buf[n:].zero_()
        shard = self._scratch("rs_ag_shard", per, t)
        dist.reduce_scatter_tensor(shard, buf, op=dist.ReduceOp.SUM, group=self.group)
        dist.all_gather_into_tensor(buf, shard, group=self.group)
        if buf is not t:
            t.copy_(buf[:n])

    # -- transport of the ordered reduce ------------------------------------
    def exchange_row_shards(self, send: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        """All-to-all of row slices.  ``send``: ``[world, R, per]`` - block ``s`` holds
        slice ``s`` of this rank's R rows.  Returns ``[world, R, per]`` whose block ``w``
        holds slice ``rank`` of rank w's rows (written into ``out`` when given: a buffer
        with a fixed address, for callers that replay recorded work on the result)."""
        if send.dim() != 3 or send.shape[0] != self.world_size or not send.is_contiguous():
            raise DeviceError("exchange_row_shards expects a contiguous [world, rows, per] tensor")
        if out is not None and (out.shape != send.shape or out.dtype != send.dtype
                                or out.device != send.device or not out.is_contiguous()):
            raise DeviceError("exchange_row_shards: out must match the send buffer")
        if self._single:
            if out is None:
                return send
            out.copy_(send)
            return out
        staged = self._host_staged(send)
        src = send.detach().to("cpu") if staged else send
        dst = out if (out is not None and not staged) else torch.empty_like(src)
        dist.all_to_all_single(dst, src, group=self.group)
        if not staged:
            return dst
        if out is None:
            return dst.to(send.device)
        out.copy_(dst)
        return out

    def allgather_shards(self, full: torch.Tensor, shard: torch.Tensor) -> torch.Tensor:
        """``full[w * per : (w + 1) * per] = rank w's shard`` on every rank, in place."""
        if full.numel() != shard.numel() * self.world_size:
            raise DeviceError("allgather_shards: the output must hold one shard per rank")
        if self._single:
            full.copy_(shard)
            return full
        if self._host_staged(full):
            host = torch.empty(full.numel(), dtype=full.dtype)
            dist.all_gather_into_tensor(host, shard.detach().to("cpu"), group=self.group)
            full.copy_(host)
        else:
            dist.all_gather_into_tensor(full, shard.contiguous(), group=self.group)
        return full

    def allgather(self, local: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        """``[world, *local.shape]`` stack of every rank's tensor (in ``out`` when given)."""
        want = (self.world_size, *local.shape)
        if out is not None and (tuple(out.shape) != want or out.dtype != local.dtype
                                or out.device != local.device or not out.is_contiguous()):
            raise DeviceError("allgather: out must be a contiguous [world, *local.shape] tensor")
        if self._single:
            if out is None:
                return local.unsqueeze(0).clone()
            out.copy_(local.unsqueeze(0))
            return out
        staged = self._host_staged(local)
        src = local.detach().to("cpu") if staged else local.contiguous()
        flat = src.reshape(-1)
        if out is not None and not staged:
            dist.all_gather_into_tensor(out.view(-1), flat, group=self.group)
            return out
        res = torch.empty(self.world_size * flat.numel(), dtype=src.dtype, device=src.device)
        dist.all_gather_into_tensor(res, flat, group=self.group)
        res = res.view(want)
        if out is not None:
            out.copy_(res)
            return out
        return res.to(local.device) if staged else res

    def max_over_ranks(self, value: float) -> float:
        if self._single:
            return float(value)
        t = torch.tensor([float(value)], dtype=torch.float64)
        if self.backend == "nccl":
            t = t.to(self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())
