"""The federated round on the device plane.

Logical pattern of the reference (nanofed/orchestration/coordinator.py:282-382
with the HTTP client loop of examples/mnist/run_experiment.py:21-86):

    broadcast global model -> every client trains locally -> clients send
    model + metrics -> server computes n_k / sum(n) weights -> weighted sum ->
    new global model -> save version + round metrics

Here every rank (one process per GPU) hosts ``G = num_clients / world``
simulated clients in one arena pool and the round is

    1. expand-copy of the global arena into the pool          (broadcast)
    2. ``GroupTrainer.train_round``                            (local training)
    3. screening of all client rows in two launches            (validation -> mask)
    4. all-gather of a 6-float record per client + one control row per rank
    5. anomaly screen, weights = mask * n / sum(mask * n), quorum flag: one launch
    6. fused weighted sum over the local rows, then one all-reduce (FedAvg)
    7. quorum select: keep the old model if too few valid updates
    8. ONE device->host copy of the gathered records           (round report)
    9. persistence on a writer thread (device->pinned copy on a side stream)

Steps 1-7 issue no host synchronisation; on a single-rank GPU run steps 3-7 are
one hipGraph replay.  Dropped, straggling, anomalous or poisoned clients are
expressed as a zero in the participation mask, because a collective cannot time
out per peer (the reference polls for a quorum of HTTP submissions instead,
coordinator.py:205-245).  A rank whose local training raised still enters every
collective of the round (with an all-invalid record block) and reports the
failure in its control row, so all ranks learn about it at step 8 and take the
recovery decision together.

Persistence keeps the reference's formats: ``ModelManager`` versions
(``models/models/*.pt`` + ``models/configs/*.json``) and
``metrics/metrics_round_<id>.json``.
"""

from __future__ import annotations

import hashlib
import json
import os
import queue
import threading
import time
from dataclasses import asdict, dataclass, field
from pathlib import Path
from typing import Callable

import numpy as np
import torch

from nanofed_amd import ops
from nanofed_amd.core.exceptions import DeviceError
from nanofed_amd.models import build_model
from nanofed_amd.parallel.comm import Communicator
from nanofed_amd.runtime.arena import ArenaLayout, ArenaPool, ParameterArena
from nanofed_amd.utils import Logger, get_current_time
from nanofed_amd.utils.streams import side_stream
from nanofed_amd.utils.trace import PhaseTimer

from .config import EngineConfig
from .data import DeviceShards, FederatedData, build_federated_data
from .trainers import (
    M_ACC,
    M_EPSILON,
    M_LOSS,
    M_SAMPLES,
    M_UPDATE_SQNORM,
    M_VALID,
    METRIC_COLUMNS,
    GroupTrainer,
    TorchGroupTrainer,
)

# exit status of a process whose watchdog saw a round overrun its time limit
# (EX_TEMPFAIL: a supervisor may start a fresh process that resumes)
WATCHDOG_EXIT_CODE = 75


class RoundFailure(RuntimeError):
    """A round failed on at least one rank; raised on EVERY rank after the
    round's collectives, so that recovery starts from a common point.
    ``local`` is this rank's own exception (None when only a peer failed)."""

    def __init__(self, message: str, local: BaseException | None, failed_ranks: list[int]) -> None:
        super().__init__(message)
        self.local = local
        self.failed_ranks = failed_ranks


@dataclass
class RoundReport:
    round_id: int
    status: str
    num_clients: int
    agg_metrics: dict[str, float]
    client_metrics: list[dict]
    start_time: str
    end_time: str
    wall_s: float
    phases_ms: dict[str, float] = field(default_factory=dict)
    # what the local trainer wants on record (e.g. which stream set it chose)
    trainer_info: dict = field(default_factory=dict)


@dataclass
class FaultPlan:
    """Fault injection for tests and drills: ``drop[r]`` = client ids that
    submit nothing in round r; ``poison[r]`` = client ids whose update is
    overwritten with NaNs before screening; ``scale[r]`` = {client id: factor
    its update (theta_k - theta) is multiplied by} (a model-poisoning client);
    ``delay[r]`` = {client id: seconds the client reports late} (left out when
    later than ``round_timeout_s``); ``fail[r]`` = message of a ``RuntimeError``
    raised once in the middle of round r, the way a HIP or RCCL error would
    surface, on every rank - or only on the ranks in ``fail_ranks`` when that is
    set; ``stall[r]`` = {rank: seconds that rank sleeps inside round r} (a hung
    peer, for watchdog drills); ``tamper[r]`` = client ids whose update is altered
    after the client signed it (one value changed on the way to the aggregator; it
    takes ``EngineConfig.sign_updates`` to notice)."""

    drop: dict[int, set[int]] = field(default_factory=dict)
    poison: dict[int, set[int]] = field(default_factory=dict)
    scale: dict[int, dict[int, float]] = field(default_factory=dict)
    delay: dict[int, dict[int, float]] = field(default_factory=dict)
    fail: dict[int, str] = field(default_factory=dict)
    fail_ranks: set[int] | None = None
    stall: dict[int, dict[int, float]] = field(default_factory=dict)
    tamper: dict[int, set[int]] = field(default_factory=dict)


def clients_of_rank(num_clients: int, world: int, rank: int) -> list[int]:
    """Contiguous block placement of clients on ranks (sizes differ by <= 1)."""
    return [int(c) for c in np.array_split(np.arange(num_clients), world)[rank]]


class _Watchdog:
    """Wall-clock limit on a round.  ``arm`` before a round, ``disarm`` after; if
    a round is still armed past its deadline the monitor thread writes a FAILED
    marker (rank 0, when a directory is known) and ends the process - a rank that
    is blocked inside a collective cannot be unblocked from the host, and a
    process that has used the GPU must not be re-executed in place, so recovery is
    a fresh process that calls ``resume()``."""

    def __init__(self, limit_s: float, first_grace_s: float, on_timeout: Callable[[int, float], None]) -> None:
        self.limit_s = limit_s
        self.first_grace_s = first_grace_s
        self._on_timeout = on_timeout
        self._lock = threading.Lock()
        self._deadline: float | None = None
        self._round = -1
        self._rounds_seen = 0
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._watch, name="nanofed-watchdog", daemon=True)
        self._thread.start()

    def arm(self, round_id: int) -> None:
        with self._lock:
            extra = self.first_grace_s if self._rounds_seen == 0 else 0.0
            self._deadline = time.monotonic() + self.limit_s + extra
            self._round = round_id
            self._rounds_seen += 1

    def disarm(self) -> None:
        with self._lock:
            self._deadline = None

    def expires_within(self, margin_s: float) -> bool:
        """Whether the armed round is past its deadline or will be within ``margin_s``."""
        with self._lock:
            return self._deadline is not None and time.monotonic() + margin_s > self._deadline

    def close(self) -> None:
        self._stop.set()

    def _watch(self) -> None:
        while not self._stop.wait(0.05):
            with self._lock:
                deadline, rnd = self._deadline, self._round
            if deadline is not None and time.monotonic() > deadline:
                self._on_timeout(rnd, self.limit_s)
                return


class _Persister:
    """Writer thread for the per-round persistence.  The main thread enqueues a
    job (host snapshot of the global arena + the report); the thread turns it into
    a model version, a metrics file and, when enabled, a checkpoint.  The queue is
    bounded by the number of snapshot buffers, so a slow disk slows the rounds
    down instead of growing memory.  An exception in the thread is kept and raised
    on the main thread at the next ``submit`` / ``flush``."""

    def __init__(self, write: Callable[[dict], None], slots: int = 2) -> None:
        self._write = write
        self._jobs: queue.Queue = queue.Queue(maxsize=slots)
        self._error: BaseException | None = None
        self.write_s = 0.0
        self.jobs_done = 0
        self._thread = threading.Thread(target=self._loop, name="nanofed-persist", daemon=True)
        self._thread.start()

    def _loop(self) -> None:
        while True:
            job = self._jobs.get()
            try:
                if job is None:
                    return
                if self._error is None:
                    t0 = time.perf_counter()
                    self._write(job)
                    self.write_s += time.perf_counter() - t0
                    self.jobs_done += 1
            except BaseException as exc:  # noqa: BLE001 - re-raised on the main thread
                self._error = exc
            finally:
                self._jobs.task_done()

    def raise_pending(self) -> None:
        if self._error is not None:
            err, self._error = self._error, None
            raise err

    def submit(self, job: dict) -> None:
        self.raise_pending()
        self._jobs.put(job)

    def flush(self) -> None:
        self._jobs.join()
        self.raise_pending()

    def close(self) -> None:
        self._jobs.join()
        self._jobs.put(None)
        # a daemon thread that is still winding down when the interpreter exits can
        # abort the process
        self._thread.join(timeout=10.0)


class FederatedEngine:
    def __init__(
        self,
        cfg: EngineConfig,
        comm: Communicator | None = None,
        data: FederatedData | None = None,
        base_dir: str | Path | None = None,
        faults: FaultPlan | None = None,
        trainer_factory: Callable[..., GroupTrainer] | None = None,
    ) -> None:
        self.cfg = cfg
        self.comm = comm or Communicator.from_env()
        self.device = self.comm.device
        self.logger = Logger()
        self.faults = faults or FaultPlan()
        self.base_dir = Path(base_dir) if base_dir is not None else None
        self.round_id = 0
        self.reports: list[RoundReport] = []
        if cfg.secure_agg not in ("none", "pairwise"):
            raise DeviceError(f"unknown secure_agg mode {cfg.secure_agg!r}")

        if cfg.num_clients < self.comm.world_size:
            raise DeviceError(
                f"{cfg.num_clients} clients cannot occupy {self.comm.world_size} ranks"
            )
        self.client_ids = clients_of_rank(
            cfg.num_clients, self.comm.world_size, self.comm.rank
        )
        counts = [
            len(clients_of_rank(cfg.num_clients, self.comm.world_size, r))
            for r in range(self.comm.world_size)
        ]
        self.max_local = max(counts)
        self._rank_counts = counts
        self._even = len(set(counts)) == 1

        torch.manual_seed(cfg.seed)  # identical initial model on every rank
        self._model_factory = lambda: build_model(cfg.model, **cfg.model_kwargs)
        self._host_model = self._model_factory()
        # (the data first: which trainer can run depends on its shape, not on its name)
        self.data = data or build_federated_data(cfg)
        self.backend = self._resolve_backend()
        if self.backend == "native" and cfg.model == "mnist_cnn":
            from .mnist_native import make_layout

            self.layout = make_layout(self._host_model)
        else:
            self.layout = ArenaLayout.from_module(self._host_model)
            if self.device.type == "cuda":
                # convolution weights live in NHWC order on the GPU (no-op for
                # models without 4-D parameters)
                self.layout = self.layout.with_channels_last()
        self.global_arena = ParameterArena(self.layout, self.device)
        self.global_arena.load_state_dict(self._host_model.state_dict())
        self.comm.broadcast(self.global_arena.data, src=0)

        self.shards = DeviceShards(
            self.data, self.client_ids, self.device, cfg.batch_size, cfg.seed,
            cfg.drop_last,
        )
        self.pool = ArenaPool(self.layout, len(self.client_ids), self.device)
        self.trainer = self._make_trainer(trainer_factory)

        K, W = cfg.num_clients, self.comm.world_size
        f32 = dict(dtype=torch.float32, device=self.device)
        self._K = K
        # rows of the round: the K simulated clients, then the slots of outside clients
        E = int(cfg.external_slots)
        if E < 0 or (E and cfg.secure_agg != "none"):
            raise DeviceError("external_slots must be >= 0 and cannot be combined with secure_agg")
        self._E, self._KT = E, K + E
        # the reduced model, in a buffer long enough for the sliced collectives to work
        # on in place (whole slices of a multiple of 64 elements per rank)
        self._partial_full = torch.zeros(self.comm.padded_numel(self.layout.numel), **f32)
        self._partial = self._partial_full[: self.layout.numel]
        self._ordered_send: torch.Tensor | None = None
        self._ordered_recv: torch.Tensor | None = None
        self._ordered_shard: torch.Tensor | None = None
        self._gathered: torch.Tensor | None = None
        self._segment_graphs: dict[str, torch.cuda.CUDAGraph] = {}
        self._tail_local: torch.Tensor | None = None
        if cfg.reduce_algo not in ("auto", "ordered", "allreduce", "rs_ag"):
            raise DeviceError(f"unknown reduce_algo {cfg.reduce_algo!r}")
        # per-client records of this rank + one control row [failed flag, ...]
        self._local_records = torch.zeros(self.max_local + 1, METRIC_COLUMNS, **f32)
        self._records = torch.zeros(K + E, METRIC_COLUMNS, **f32)
        self._weights = torch.zeros(K + E, **f32)
        self._ext_full: torch.Tensor | None = None
        if E:
            # [E, padded numel] so that the sliced reduce can cut them like the arena
            self._ext_full = torch.zeros(E, self.comm.padded_numel(self.layout.numel), **f32)
            self._ext_rows = self._ext_full[:, : self.layout.numel]
            self._ext_records = torch.zeros(E, METRIC_COLUMNS, **f32)
            self._ext_stage = ParameterArena(self.layout, "cpu")
            self._ext_pending: dict[int, tuple[torch.Tensor, list[float]]] = {}
        self._ext_names = [f"external_{e}" for e in range(E)]
        self._flags = torch.zeros(4, **f32)
        self._control = torch.zeros(W, **f32)
        # everything the host reads at the end of a round, in one buffer
        self._host_pack = torch.zeros((K + E) * METRIC_COLUMNS + 4 + W, **f32)
        # the next round's broadcast and local training, queued behind this round's tail
        # before the host looks at this round's outcome: (round id, metrics tensor)
        self._prelaunched: tuple[int, torch.Tensor] | None = None
        self._host_pack_pinned: torch.Tensor | None = None
        self._pack_event = None
        self._select_done = None
        if self.device.type == "cuda":
            self._host_pack_pinned = torch.empty_like(self._host_pack, device="cpu").pin_memory()
            self._pack_event = torch.cuda.Event()
            self._select_done = torch.cuda.Event()
        self._need_delta = cfg.max_update_norm > 0 or cfg.zscore_threshold > 0
        self._tail_graph: torch.cuda.CUDAGraph | None = None
        self._tail_rounds = 0
        self._own_seed_rows: torch.Tensor | None = None
        self.secure_public_keys: list[bytes] = []
        self._shares: torch.Tensor | None = None
        if cfg.secure_agg == "pairwise":
            self._setup_secure_agg()
        # signed updates: one key pair per simulated client of this rank; the public
        # halves are what the aggregating side verifies against
        self._signers: list = []
        self.client_public_keys: dict[int, bytes] = {}
        self.unverified_updates = 0
        if cfg.sign_updates:
            from nanofed_amd.crypto.backend import RSAKey

            self._signers = [RSAKey.generate(2048) for _ in self.client_ids]
            self.client_public_keys = {
                cid: key.public_pem() for cid, key in zip(self.client_ids, self._signers)
            }

        self._model_manager = None
        self._recovery = None
        self.recoveries = 0
        self._resumed = False
        self._stale_checked = False
        self._timer = PhaseTimer(self.device)
        self._persister: _Persister | None = None
        self._snap_slots: list[torch.Tensor] = []
        self._snap_events: list = []
        self._snap_next = 0
        self._copy_stream = None
        # recovery point of last resort (no completed checkpoint yet): the global
        # model as it was when the failed round began
        self._round_start = (
            torch.empty_like(self.global_arena.data) if cfg.checkpoint_rounds else None
        )
        if self.base_dir is not None and self.comm.is_root:
            self._setup_persistence()
        self._in_run = False
        self._more_rounds = False  # inside run(): another round follows the current one
        self._watchdog: _Watchdog | None = None
        if cfg.enforce_round_timeout and cfg.round_timeout_s > 0:
            self._watchdog = _Watchdog(
                cfg.round_timeout_s, cfg.first_round_grace_s, self._on_round_timeout
            )

    # -- construction helpers ---------------------------------------------
    def _make_trainer(self, factory: Callable[..., GroupTrainer] | None) -> GroupTrainer:
        cfg = self.cfg
        if factory is not None:
            return factory(cfg, self.pool, self.shards, self._model_factory, self.data)
        if self.backend == "native" and cfg.model == "mnist_cnn":
            from .mnist_native import MnistCnnNativeTrainer

            return MnistCnnNativeTrainer(cfg, self.pool, self.shards, self.data)
        if self.backend == "native" and cfg.model.startswith("gpt2"):
            from .gpt2_native import GPT2NativeTrainer

            return GPT2NativeTrainer(cfg, self.pool, self.shards, self.data, self._host_model.cfg)
        if self.backend == "native" and cfg.model == "resnet18":
            from .resnet_native import ResNetNativeTrainer

            return ResNetNativeTrainer(
                cfg, self.pool, self.shards, self._model_factory,
                self.data.eval_inputs, self.data.eval_labels,
            )
        if self.backend == "native":
            raise DeviceError(f"no native trainer for model {cfg.model!r}")
        return TorchGroupTrainer(
            cfg, self.pool, self.shards, self._model_factory,
            self.data.eval_inputs, self.data.eval_labels,
        )

    def _resolve_backend(self) -> str:
        cfg = self.cfg
        if cfg.backend != "auto":
            return cfg.backend
        on_gpu = self.device.type == "cuda"
        mnist_ok = False
        if cfg.model == "mnist_cnn" and on_gpu:
            from .mnist_native import supports

            mnist_ok = supports(cfg)
        gpt2_ok = False
        if cfg.model.startswith("gpt2") and on_gpu and cfg.dp is None and cfg.dtype == "bf16":
            m = self._host_model.cfg
            gpt2_ok = (
                m.n_embd == 64 * m.n_head and m.dropout == 0.0 and cfg.seq_len % 128 == 0
                and m.n_embd <= 2048
            )
        resnet_ok = False
        if cfg.model == "resnet18" and on_gpu:
            from .resnet_native import supports as resnet_supports

            resnet_ok = resnet_supports(cfg, self._host_model, self.device, self.data)
        return "native" if (mnist_ok or gpt2_ok or resnet_ok) else "torch"

    def _setup_persistence(self) -> None:
        from nanofed_amd.server.model_manager import ModelManager

        base = self.base_dir
        for sub in ("metrics", "data", "models/configs", "models/models"):
            (base / sub).mkdir(parents=True, exist_ok=True)
        self._model_manager = ModelManager(self._host_model)
        self._model_manager.set_dirs(base / "models" / "models", base / "models" / "configs")
        if self.cfg.checkpoint_rounds:
            from nanofed_amd.server.fault_tolerance import FaultTolerantCoordinator

            self._recovery = FaultTolerantCoordinator(base)
        n_slots = 2
        pin = self.device.type == "cuda"
        self._snap_slots = [
            torch.empty(self.layout.numel, dtype=torch.float32, pin_memory=pin)
            for _ in range(n_slots)
        ]
        if pin:
            self._copy_stream = side_stream(self.device)
            self._snap_events = [torch.cuda.Event() for _ in range(n_slots)]
        if self.cfg.async_persist:
            self._persister = _Persister(self._write_job, slots=n_slots - 1)

    def enable_persistence(self, base_dir: str | Path) -> None:
        """Start writing model versions / metrics / checkpoints under
        ``base_dir`` from the next round on (rank 0 only writes)."""
        self.flush()
        self.base_dir = Path(base_dir)
        if self.comm.is_root:
            self._setup_persistence()

    def disable_persistence(self) -> None:
        """Stop writing per-round files (the files written so far stay)."""
        self.flush()
        if self._persister is not None:
            self._persister.close()
            self._persister = None
        self._model_manager = None
        self._recovery = None

    def _setup_secure_agg(self) -> None:
        """Pair seeds of the blinding masks, by key agreement: every client draws an X25519
        key pair, the public halves travel in one all-gather, and client i derives the seed
        it shares with peer j as SHA-256(X25519(sk_i, pk_j) || context) - the same 63 bits
        on both sides, known to nobody else.  The aggregating side of this object keeps the
        public keys only; a client's seeds stay in its own ``PeerMaskPlan`` and its row of
        ``_own_seed_rows`` (from which a survivor reveals, after the reduce, the seeds it
        shared with clients that dropped out - what the repair of their unmatched masks
        needs; reference capability: nanofed/server/aggregator/secure.py:192-313)."""
        from nanofed_amd.crypto.backend import X25519Key

        cfg, K = self.cfg, self.cfg.num_clients
        if K < 2:
            raise DeviceError("pairwise secure aggregation needs at least two clients")
        keys: dict[int, X25519Key] = {}
        for cid in self.client_ids:
            if cfg.secure_key_seed is None:
                keys[cid] = X25519Key.generate()
            else:
                keys[cid] = X25519Key(hashlib.sha256(
                    f"nanofed-x25519|{cfg.secure_key_seed}|{cid}".encode()).digest())
        mine = torch.zeros(self.max_local, 32, dtype=torch.uint8)
        for k, cid in enumerate(self.client_ids):
            mine[k] = torch.frombuffer(bytearray(keys[cid].public), dtype=torch.uint8)
        gathered = self.comm.allgather(mine.to(self.device)).cpu()  # [W, max_local, 32]
        public: list[bytes] = []
        for w, count in enumerate(self._rank_counts):
            public += [bytes(gathered[w, i].tolist()) for i in range(count)]
        if len(set(public)) != K:
            raise DeviceError("secure aggregation: two clients announced the same public key")
        self.secure_public_keys = public
        own = torch.zeros(K, K, dtype=torch.int64)
        self._mask_plans: list[ops.PeerMaskPlan] = []
        for cid in self.client_ids:
            peers = [j for j in range(K) if j != cid]
            seeds = []
            for j in peers:
                context = b"nanofed-pair-mask|%d|%d" % (min(cid, j), max(cid, j))
                digest = keys[cid].derive(public[j], context)
                seeds.append(int.from_bytes(digest[:8], "big") >> 1)
                own[cid, j] = seeds[-1]
            self._mask_plans.append(
                ops.PeerMaskPlan(seeds, [1.0 if cid < j else -1.0 for j in peers]))
        # (the private keys are dropped here: the seeds are all a client needs afterwards)
        self._own_seed_rows = own.to(self.device)
        # a client's share of the total sample count is public metadata (the
        # counts travel in the clear in the reference too, client.py:180-189)
        sizes = torch.tensor([float(len(s)) for s in self.data.shards], dtype=torch.float32)
        self._shares_host = (sizes / sizes.sum().clamp(min=1.0)).tolist()
        self._shares = torch.tensor(self._shares_host, dtype=torch.float32).to(self.device)

    # -- mixed rounds ---------------------------------------------------------------------
    def submit_external(self, slot: int, state, num_samples: float,
                        metrics: dict | None = None, client_id: str | None = None) -> None:
        """Hand in the update of a client that trains outside this engine, for the round
        that runs next (reference: a round is made of the updates that were POSTed,
        nanofed/orchestration/coordinator.py:205-245,307-322).  ``state`` is a
        ``state_dict`` of the model or a flat tensor in arena order; the update becomes row
        ``K + slot`` of the round: screened like the resident clients' rows (finite values,
        norm bound, z-score), weighted by ``num_samples``, added in the same fused sum.
        With several ranks only rank 0 takes submissions; it broadcasts them in the round.
        A slot nobody filled for a round takes no part in it."""
        if not 0 <= slot < self._E:
            raise DeviceError(f"external slot {slot} out of range (external_slots={self._E})")
        if not self.comm.is_root:
            raise DeviceError("external updates are submitted on rank 0")
        if isinstance(state, torch.Tensor):
            if state.numel() != self.layout.numel:
                raise DeviceError("a flat external update must have the arena's length")
            flat = state.detach().reshape(-1).to("cpu", torch.float32).clone()
        else:
            self._ext_stage.load_state_dict(state)
            flat = self._ext_stage.data.clone()
        m = metrics or {}
        record = [0.0] * METRIC_COLUMNS
        record[M_SAMPLES] = float(num_samples)
        record[M_LOSS] = float(m.get("loss", 0.0))
        record[M_ACC] = float(m.get("accuracy", 0.0))
        record[M_VALID] = 1.0 if num_samples > 0 else 0.0
        self._ext_pending[slot] = (flat, record)
        if client_id is not None:
            self._ext_names[slot] = str(client_id)

    def _load_external(self) -> None:
        """Pending submissions -> the external rows and their records (rank 0), then to
        every rank.  Eager, host-driven: a round with external slots is not graph-captured."""
        rec = torch.zeros(self._E, METRIC_COLUMNS, dtype=torch.float32)
        if self.comm.is_root:
            for slot, (flat, record) in self._ext_pending.items():
                self._ext_rows[slot].copy_(flat)
                rec[slot] = torch.tensor(record)
            self._ext_pending.clear()
        self._ext_records.copy_(rec)
        if self.comm.collective:
            self.comm.broadcast(self._ext_records, src=0)
            self.comm.broadcast(self._ext_full.view(-1), src=0)

    # -- the round --------------------------------------------------------
    def run_round(self, flush: bool | None = None, prefetch_next: bool | None = None) -> RoundReport:
        """One federated round.  ``flush=False`` leaves the round's persistence
        with the writer thread (``run`` and the benchmark flush once at the end);
        the default flushes unless the call comes from ``run``.

        ``prefetch_next=True`` says that the caller will run another round right after
        this one: its broadcast and local training are then queued on the device behind
        this round's tail *before* the host waits for this round's outcome, so the device
        does not idle while the host builds the report and hands the round to the writer
        (``run`` does this for every round but the last).  If this round does not complete
        (quorum lost), the queued work is for a round that will not come; the next call
        notices the round number and trains again from the unchanged model - the trainers
        keep no state across rounds, so nothing has to be undone."""
        if flush is None:
            flush = not self._in_run
        if prefetch_next is None:
            prefetch_next = self._in_run and self._more_rounds
        r = self.round_id
        t0 = time.perf_counter()
        start = get_current_time()
        if self._watchdog is not None:
            self._watchdog.arm(r)
        try:
            report = self._round(r, t0, start, prefetch_next and self._may_prefetch())
        except Exception:
            # A peer whose watchdog fired a moment before ours has left the job, and its
            # exit surfaces here as a failed collective: that is the same time-out, seen
            # from the other side - end the same way (FAILED record, exit code for the
            # supervisor) instead of racing our own watchdog with a traceback.
            if self._watchdog is not None and self._watchdog.expires_within(
                min(1.0, max(0.2, 0.1 * self._watchdog.limit_s))
            ):
                self._on_round_timeout(r, self._watchdog.limit_s)
            raise
        finally:
            if self._watchdog is not None:
                self._watchdog.disarm()
        if flush:
            self.flush()
        return report

    def _may_prefetch(self) -> bool:
        """Queueing the next round early is kept to runs whose rounds the host does not
        touch between training and aggregation and that do not roll back: no fault plan,
        no signatures, no checkpoint-and-recover (its recovery point of last resort is the
        model at the start of the *failed* round)."""
        cfg = self.cfg
        f = self.faults
        planned = any((f.drop, f.poison, f.scale, f.delay, f.fail, f.stall, f.tamper))
        return (
            cfg.prefetch_rounds and not planned and not self._signers
            and not cfg.checkpoint_rounds and cfg.secure_agg != "pairwise"
        )

    def _round(self, r: int, t0: float, start, prefetch_next: bool = False) -> RoundReport:
        cfg, comm = self.cfg, self.comm
        timer = self._timer
        multi = comm.world_size > 1
        failure: BaseException | None = None
        # a persistence error of an earlier round (rank 0) becomes this round's failure
        if self._persister is not None:
            try:
                self._persister.raise_pending()
            except Exception as exc:  # noqa: BLE001
                failure = exc
        # this round's broadcast and training may already be in the queue (run_round)
        queued, self._prelaunched = self._prelaunched, None
        if queued is not None and (queued[0] != r or failure is not None):
            queued = None  # for a round that did not come: train again from the model as it is
        # 1. broadcast
        if queued is None:
            with timer.phase("broadcast"):
                self.pool.broadcast_from(self.global_arena.data)
                if self._round_start is not None:
                    self._round_start.copy_(self.global_arena.data)
        # 2. local training
        local = None
        try:
            if failure is None:
                if queued is not None:
                    local = queued[1]
                else:
                    with timer.phase("local_train"):
                        local = self.trainer.train_round(r)  # [G, METRIC_COLUMNS] on device
                signatures = self._sign_rows(r) if self._signers else None
                self._inject_faults(r, local)
                if signatures is not None:
                    with timer.phase("verify_signatures"):
                        self._verify_rows(r, signatures, local)
        except Exception as exc:  # noqa: BLE001 - agreed on below, re-raised on every rank
            failure = exc
        if failure is not None and not multi:
            raise failure
        self._tail(local, failed=failure is not None, eager=failure is not None or self._faulty(r))
        if self._select_done is not None:
            self._select_done.record()  # what a snapshot of the new model waits for
        # 8. the only host synchronisation of the round
        spans = None
        if prefetch_next and failure is None:
            # copy the outcome out without blocking, queue the next round behind it, and
            # only then wait for the copy
            if self._host_pack_pinned is not None:
                self._host_pack_pinned.copy_(self._host_pack, non_blocking=True)
                self._pack_event.record()
                host = self._host_pack_pinned
            else:
                host = self._host_pack.clone()
            spans = timer.detach()  # this round's phases; the queued round's go to the next
            try:
                with timer.phase("broadcast"):
                    self.pool.broadcast_from(self.global_arena.data)
                with timer.phase("local_train"):
                    self._prelaunched = (r + 1, self.trainer.train_round(r + 1))
            except Exception:  # noqa: BLE001 - the round itself will run into it again
                self._prelaunched = None
                timer.detach()
            if self._host_pack_pinned is not None:
                self._pack_event.synchronize()
        else:
            host = self._host_pack.cpu()
        K, W = self._KT, comm.world_size
        rec = host[: K * METRIC_COLUMNS].view(K, METRIC_COLUMNS).numpy()
        flags = host[K * METRIC_COLUMNS : K * METRIC_COLUMNS + 4]
        control = host[K * METRIC_COLUMNS + 4 :]
        if bool((control > 0).any()):
            ranks = [w for w in range(W) if control[w] > 0]
            raise RoundFailure(
                f"round {r} failed on rank(s) {ranks}"
                + (f": {type(failure).__name__}: {failure}" if failure is not None else ""),
                failure, ranks,
            ) from failure
        succeeded = bool(flags[0] > 0)
        if spans is not None:
            phases = timer.summarize(spans) if cfg.phase_timing else {}
        else:
            phases = timer.collect() if cfg.phase_timing else {}

        report = self._report(r, rec, succeeded, start, time.perf_counter() - t0)
        report.phases_ms = phases
        report.trainer_info = dict(getattr(self.trainer, "stream_report", None) or {})
        if flags[3] > 0:
            report.agg_metrics["anomalous_clients"] = float(flags[3])
        self.reports.append(report)
        if succeeded:
            self.round_id += 1
            t_ck = time.perf_counter()
            self._persist(report)
            if self._model_manager is not None:
                report.phases_ms["checkpoint"] = (time.perf_counter() - t_ck) * 1000.0
        return report

    def _faulty(self, r: int) -> bool:
        f = self.faults
        return any(r in d for d in (f.drop, f.poison, f.scale, f.delay, f.fail, f.stall, f.tamper))

    def _inject_faults(self, r: int, local: torch.Tensor) -> None:
        cfg = self.cfg
        f = self.faults
        if r in f.stall and self.comm.rank in f.stall[r]:
            time.sleep(f.stall[r][self.comm.rank])
        if r in f.fail and (f.fail_ranks is None or self.comm.rank in f.fail_ranks):
            raise RuntimeError(f.fail.pop(r))
        late = {
            cid for cid, secs in f.delay.get(r, {}).items()
            if cfg.round_timeout_s > 0 and secs > cfg.round_timeout_s
        }
        for k, cid in enumerate(self.client_ids):
            if cid in f.scale.get(r, {}):
                row = self.pool[k].data
                g = self.global_arena.data
                row.copy_(g + (row - g) * float(f.scale[r][cid]))
            if cid in f.poison.get(r, ()):
                self.pool[k].data.fill_(float("nan"))
            if cid in f.tamper.get(r, ()):
                self.pool[k].data[7] += 0.125  # one value altered after signing
            if cid in f.drop.get(r, ()) or cid in late:
                local[k, M_VALID] = 0.0

    # -- signed updates ---------------------------------------------------------------
    @staticmethod
    def _signing_message(r: int, cid: int, digest: bytes) -> bytes:
        return b"nanofed-amd/update/v1|" + r.to_bytes(8, "big") + cid.to_bytes(8, "big") + digest

    def _sign_rows(self, r: int) -> list[bytes]:
        """Client side: every local client signs (round, its id, the tree digest of its
        arena row) with its private key (RSA-PSS, SHA-256).  Reference:
        SecurityManager.sign_update, nanofed/server/validation.py:189-194 - there over the
        serialised tensors, here over a digest computed where the row lies."""
        from nanofed_amd.ops.digest_ops import row_digests

        digests = row_digests(self.pool.data)
        return [
            key.pss_sign(self._signing_message(r, cid, d))
            for key, cid, d in zip(self._signers, self.client_ids, digests)
        ]

    def _verify_rows(self, r: int, signatures: list[bytes], local: torch.Tensor) -> None:
        """Aggregating side: recompute the digest of every row as it arrived and check it
        against the client's signature and registered public key; a row that fails takes
        no part in the round (nanofed/server/validation.py:196-207)."""
        from nanofed_amd.crypto.backend import RSAKey
        from nanofed_amd.ops.digest_ops import row_digests

        digests = row_digests(self.pool.data)
        bad = []
        for k, (cid, d, sig) in enumerate(zip(self.client_ids, digests, signatures)):
            key = RSAKey.from_public_pem(self.client_public_keys[cid])
            if not key.pss_verify(self._signing_message(r, cid, d), sig):
                bad.append(k)
        if bad:
            self.unverified_updates += len(bad)
            self.logger.warning(
                f"round {r}: signature check failed for clients "
                f"{[self.client_ids[k] for k in bad]}; their updates are left out"
            )
            local[torch.tensor(bad, device=local.device), M_VALID] = 0.0

    # steps 3-7
    def _tail(self, local: torch.Tensor | None, failed: bool, eager: bool) -> None:
        cfg = self.cfg
        quiet = (
            cfg.tail_graph and self.device.type == "cuda" and not eager and local is not None
            # (the blinding masks take the round number as their stream offset: a kernel
            # argument, frozen in a recorded graph)
            and cfg.secure_agg == "none" and self._E == 0
        )
        if quiet and not self.comm.collective:
            if self._tail_graph is None and self._tail_rounds >= 1:
                # captured on the second eager-free round: every buffer exists and the
                # trainer's metrics tensor has its final address
                self._tail_local = local
                torch.cuda.synchronize(self.device)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    self._tail_ops(local, failed=False)
                self._tail_graph = graph
            if self._tail_graph is not None and local.data_ptr() == self._tail_local.data_ptr():
                with self._timer.phase("tail_graph"):
                    self._tail_graph.replay()
                self._tail_rounds += 1
                return
        # Several ranks: the collectives stay eager (no library call is recorded), the
        # device work between them is replayed from one graph per segment - recorded on
        # the second quiet round, when every buffer has its final address.
        segments = quiet and self.comm.collective and self._tail_rounds >= 1
        if segments and self._tail_local is None:
            self._tail_local = local
        if segments and local.data_ptr() != self._tail_local.data_ptr():
            segments = False
        self._tail_ops(local, failed, self._timer, segments)
        self._tail_rounds += 1

    def _segment(self, key: str, fn: Callable[[], None], replayable: bool) -> None:
        """Run ``fn`` - eagerly, or (``replayable``) as a hipGraph recorded at its first
        replayable call.  ``fn`` must touch only buffers whose addresses never change."""
        if not replayable:
            fn()
            return
        graph = self._segment_graphs.get(key)
        if graph is None:
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                fn()
            self._segment_graphs[key] = graph
        graph.replay()

    def _tail_ops(self, local: torch.Tensor | None, failed: bool, timer: PhaseTimer | None = None,
                  segments: bool = False) -> None:
        """Steps 3-7.  ``timer`` is None while the steps are being recorded into a
        graph (events cannot be taken inside a capture).  ``segments``: replay the device
        work between the collectives from graphs (see ``_tail``)."""
        from contextlib import nullcontext

        cfg, comm = self.cfg, self.comm
        G, K, M = len(self.client_ids), self._K, METRIC_COLUMNS
        rec = self._local_records

        def screen() -> None:
            rec.zero_()
            if local is not None and not failed:
                rec[:G].copy_(local)
                # 3. screening -> participation mask
                if cfg.validate_updates:
                    ops.screen_rows(
                        self.pool.data, rec, M_UPDATE_SQNORM, M_VALID,
                        ref=self.global_arena.data if self._need_delta else None,
                        max_norm=cfg.max_update_norm,
                    )
            if failed:
                rec[self.max_local, 0] = 1.0

        def assemble() -> torch.Tensor:
            # 4b. the K records in client order (+ the ranks' control flags)
            if not comm.collective:
                self._control.copy_(rec[self.max_local, :1])
                return rec[:K]
            gathered = self._gathered  # [W, max_local + 1, M]
            if self._even:
                self._records[:K].copy_(gathered[:, : self.max_local].reshape(K, M))
            else:
                self._records[:K].copy_(torch.cat(
                    [gathered[w, :c] for w, c in enumerate(self._rank_counts)], dim=0
                ))
            self._control.copy_(gathered[:, self.max_local, 0])
            return self._records

        with timer.phase("screen_gather") if timer is not None else nullcontext():
            self._segment("screen", screen, segments)
            # 4a. gather the per-client records (+ control rows)
            if comm.collective:
                if self._gathered is None:
                    self._gathered = torch.zeros(comm.world_size, *rec.shape, dtype=rec.dtype,
                                                 device=rec.device)
                comm.allgather(rec, out=self._gathered)
        with timer.phase("reduce") if timer is not None else nullcontext():
            if cfg.secure_agg == "pairwise" or self._E:
                records = assemble()
                if self._E:
                    records = self._append_external(records)
                self._reduce_select(records, G, self._KT, M)
                return
            # 5-7 in segments around the collectives of the reduce
            algo = self._reduce_algo()
            ordered = comm.collective and algo == "ordered"
            lo = self.client_ids[0]

            def pre() -> None:
                records = assemble()
                self._weigh(records)
                if ordered:
                    self._ordered_fill_send(G)
                else:
                    ops.weighted_sum(self.pool.rows(), self._weights[lo : lo + G], out=self._partial)
                # (the records are final here: into the host's buffer with them)
                self._host_pack[: K * M].copy_(records.reshape(-1))

            def mid() -> None:
                ops.weighted_sum(self._ordered_sources(self._ordered_recv), self._weights,
                                 out=self._ordered_shard)

            def post() -> None:
                # 7. quorum select (device-side; the old model survives a failed round)
                ops.select_copy(self.global_arena.data, self._partial, self._flags)
                self._host_pack[K * M : K * M + 4].copy_(self._flags)
                self._host_pack[K * M + 4 :].copy_(self._control)

            self._segment("pre", pre, segments)
            if ordered:
                comm.exchange_row_shards(self._ordered_send, out=self._ordered_recv)
                self._segment("mid", mid, segments)
                comm.allgather_shards(self._partial_full, self._ordered_shard)
            elif comm.collective:
                comm.allreduce_sum(self._partial_full, algo=algo)
            self._segment("post", post, segments)

    def _append_external(self, records: torch.Tensor) -> torch.Tensor:
        """The rows of outside clients: loaded, screened like the others, appended."""
        cfg, K = self.cfg, self._K
        self._load_external()
        if cfg.validate_updates:
            ops.screen_rows(
                self._ext_rows, self._ext_records, M_UPDATE_SQNORM, M_VALID,
                ref=self.global_arena.data if self._need_delta else None,
                max_norm=cfg.max_update_norm,
            )
        if records is not self._records:
            self._records[:K].copy_(records)
        self._records[K:].copy_(self._ext_records)
        return self._records

    def _weigh(self, records: torch.Tensor) -> None:
        """5. anomaly screen, weights, quorum flag: one launch."""
        cfg = self.cfg
        ops.round_weights(
            records, M_SAMPLES, M_VALID, M_UPDATE_SQNORM, float(cfg.required_clients()),
            self._weights, self._flags, z_threshold=cfg.zscore_threshold,
            min_clients_for_stats=cfg.min_clients_for_stats,
        )

    def _reduce_select(self, records: torch.Tensor, G: int, K: int, M: int) -> None:
        """Steps 5-7 in one piece (rounds under pairwise masking or with outside rows)."""
        cfg = self.cfg
        self._weigh(records)
        lo = self.client_ids[0]
        # 6. FedAvg: fused local weighted sum + one all-reduce (rows with weight 0
        #    are never read, so a poisoned row cannot leak NaNs)
        if cfg.secure_agg == "pairwise":
            self._secure_reduce(records, lo, G)
        else:
            self._reduce_rows(self._weights, lo, G)
        # 7. quorum select (device-side; the old model survives a failed round)
        ops.select_copy(self.global_arena.data, self._partial, self._flags)
        pack = self._host_pack
        pack[: K * M].copy_(records.reshape(-1))
        pack[K * M : K * M + 4].copy_(self._flags)
        pack[K * M + 4 :].copy_(self._control)

    def _reduce_algo(self) -> str:
        algo = self.cfg.reduce_algo
        if algo == "auto":
            from nanofed_amd.parallel.comm import RS_AG_THRESHOLD_BYTES

            small = self.layout.numel * 4 < RS_AG_THRESHOLD_BYTES
            algo = "ordered" if small else "rs_ag"
        return algo

    def _reduce_rows(self, weights: torch.Tensor, lo: int, G: int) -> None:
        """``self._partial = sum_k weights[k] * row_k`` over the K client rows of all
        ranks; ``weights`` is the device vector of all K weights, this rank's rows are
        clients ``lo .. lo + G - 1``.

        *ordered*: every rank receives slice ``rank`` of every client row, adds the K
        slices in client order - the very chain of fused multiply-adds that one process
        runs over its K rows, so the sum has the same bits for any number of ranks - and
        the reduced slices are gathered.  Rows with weight 0 are never read.
        Otherwise: fused weighted sum of the local rows, then the library's all-reduce."""
        comm = self.comm
        algo = self._reduce_algo()
        K = self._K
        if not comm.collective or algo != "ordered":
            rows, w = self.pool.rows(), weights[lo : lo + G]
            if self._E and comm.is_root:
                # (one rank adds the outside rows; in one process the chain of additions is
                # the one a K + E client engine runs)
                rows = rows + [self._ext_rows[e] for e in range(self._E)]
                w = torch.cat([w, weights[K:]])
            ops.weighted_sum(rows, w, out=self._partial)
            if comm.collective:
                comm.allreduce_sum(self._partial_full, algo=algo)
            return
        self._ordered_fill_send(G)
        comm.exchange_row_shards(self._ordered_send, out=self._ordered_recv)
        ops.weighted_sum(self._ordered_sources(self._ordered_recv), weights, out=self._ordered_shard)
        comm.allgather_shards(self._partial_full, self._ordered_shard)

    def _ordered_fill_send(self, G: int) -> None:
        """Block s of the send buffer = slice s of this rank's rows (rows beyond G and the
        tail of the last slice stay zero: they carry weight 0 / are cut off)."""
        comm = self.comm
        W, R, n = comm.world_size, self.max_local, self.layout.numel
        per = comm.shard_numel(n)
        if self._ordered_send is None:
            f32 = dict(dtype=torch.float32, device=self.device)
            self._ordered_send = torch.zeros(W, R, per, **f32)
            self._ordered_recv = torch.zeros(W, R, per, **f32)
            self._ordered_shard = torch.zeros(per, **f32)
        send = self._ordered_send
        rows = self.pool.data  # [G, n]
        whole = n // per  # slices that lie completely inside a row
        if whole:
            send[:whole, :G].copy_(rows[:, : whole * per].view(G, whole, per).transpose(0, 1))
        if whole < W and n > whole * per:
            send[whole, :G, : n - whole * per].copy_(rows[:, whole * per :])

    def _ordered_sources(self, recv: torch.Tensor) -> list[torch.Tensor]:
        """Slice ``rank`` of every client row, in client order (= rank-major order of the
        block placement), followed by this rank's slice of the outside rows."""
        comm = self.comm
        per = comm.shard_numel(self.layout.numel)
        sources = [recv[w, i] for w, count in enumerate(self._rank_counts) for i in range(count)]
        assert len(sources) == self._K
        for e in range(self._E):
            sources.append(self._ext_full[e, comm.rank * per : (comm.rank + 1) * per])
        return sources

    def _secure_reduce(self, records: torch.Tensor, lo: int, G: int) -> None:
        """FedAvg under pairwise masking (reference capability: the share sums of
        nanofed/server/aggregator/privacy.py:72-110 and the masked sums of
        secure.py:192-313).  Client k submits share_k * theta_k blinded with one
        seeded mask per peer (+ towards larger ids, - towards smaller ones), where
        share_k = n_k / sum of all registered n.  The sum over the present
        clients is taken with weight 1, the masks a missing client left without
        partner are removed from the reduced sum on the device, and the result is
        divided by the sum of the present shares."""
        cfg, comm = self.cfg, self.comm
        scale = cfg.secure_mask_scale
        offset = self.round_id
        for k, cid in enumerate(self.client_ids):
            ops.pairwise_mask_(self.pool[k].data, self._mask_plans[k], None, scale,
                               prescale=self._shares_host[cid], offset=offset)
        present = (records[:, M_VALID] > 0).to(torch.float32)
        self._reduce_rows(present, lo, G)
        # The survivors reveal the seeds they shared with the clients that are missing:
        # entry (i, j) of the table is filled by the rank of a present client i for a
        # missing client j, and nothing else is - decided on the device, no host round trip.
        here = present > 0
        reveal = self._own_seed_rows * (here[:, None] & ~here[None, :]).to(torch.int64)
        reveal = reveal + reveal.t()
        if comm.collective:
            comm.allreduce_sum(reveal.view(-1), algo="allreduce")
        ops.pairwise_repair_(self._partial, present, reveal.contiguous(), scale, offset=offset)
        mass = (present * self._shares).sum().clamp(min=1e-30)
        self._partial.div_(mass)

    # -- reporting / persistence ------------------------------------------
    def _report(
        self, r: int, rec: np.ndarray, ok: bool, start, wall: float
    ) -> RoundReport:
        mask = rec[:, M_VALID] > 0
        mass = rec[:, M_SAMPLES] * mask
        tot = float(mass.sum()) or 1.0
        w = mass / tot
        agg = {
            "loss": float((w * np.nan_to_num(rec[:, M_LOSS])).sum()),
            "accuracy": float((w * np.nan_to_num(rec[:, M_ACC])).sum()),
            "samples_processed": float((w * rec[:, M_SAMPLES]).sum()),
        }
        if self.cfg.dp is not None:
            agg["privacy_epsilon"] = float(rec[:, M_EPSILON].max())
        clients = [
            {
                "client_id": f"client_{cid}" if cid < self._K else self._ext_names[cid - self._K],
                "metrics": {
                    "loss": float(rec[cid, M_LOSS]),
                    "accuracy": float(rec[cid, M_ACC]),
                    "samples_processed": int(rec[cid, M_SAMPLES]),
                },
                "weight": float(w[cid]),
                "valid": bool(mask[cid]),
                "update_norm": float(np.sqrt(max(rec[cid, M_UPDATE_SQNORM], 0.0))),
            }
            for cid in range(rec.shape[0])
        ]
        return RoundReport(
            round_id=r,
            status="COMPLETED" if ok else "FAILED",
            num_clients=int(mask.sum()),
            agg_metrics=agg,
            client_metrics=clients,
            start_time=start.isoformat(),
            end_time=get_current_time().isoformat(),
            wall_s=wall,
        )

    def _snapshot(self) -> tuple[torch.Tensor, object]:
        """Host copy of the global arena for the writer: a device->pinned copy on
        the side stream, ordered after the round's select by an event; the training
        stream does not wait for it (the next round only reads the arena)."""
        slot = self._snap_next
        self._snap_next = (slot + 1) % len(self._snap_slots)
        buf = self._snap_slots[slot]
        if self._copy_stream is None:
            buf.copy_(self.global_arena.data)
            return buf, None
        main = torch.cuda.current_stream(self.device)
        # after the round's select; not after whatever else is queued by now (the next
        # round's training, which only reads the arena)
        self._copy_stream.wait_event(self._select_done)
        with torch.cuda.stream(self._copy_stream):
            buf.copy_(self.global_arena.data, non_blocking=True)
            self._snap_events[slot].record()
        # the arena must not be overwritten (next round's select) before the copy ran
        main.wait_event(self._snap_events[slot])
        return buf, self._snap_events[slot]

    def _persist(self, report: RoundReport) -> None:
        if self._model_manager is None:
            return
        self._retire_stale_checkpoints()
        if self._persister is not None:
            # a free snapshot buffer is guaranteed by the bounded queue: wait for
            # the writer before reusing the oldest one
            while self._persister._jobs.full():
                self._persister._jobs.join()
        buf, event = self._snapshot()
        job = {
            "report": report, "flat": buf, "event": event, "next_round": self.round_id,
            "phases_ms": dict(report.phases_ms),
        }
        if self._persister is not None:
            self._persister.submit(job)
        else:
            self._write_job(job)

    def _write_job(self, job: dict) -> None:
        report: RoundReport = job["report"]
        if job["event"] is not None:
            job["event"].synchronize()
        state = ParameterArena(self.layout, data=job["flat"]).state_dict("cpu")
        self._host_model.load_state_dict(state)
        self._model_manager.save_model(
            config={
                "round_id": report.round_id,
                "client_metrics": report.client_metrics,
                "client_weights": {
                    c["client_id"]: c["weight"] for c in report.client_metrics
                },
                "start_time": report.start_time,
                "status": "AGGREGATING",
                "num_clients": report.num_clients,
            },
            metrics=report.agg_metrics,
        )
        path = self.base_dir / "metrics" / f"metrics_round_{report.round_id}.json"
        tmp = path.with_suffix(".json.tmp")
        with tmp.open("w") as f:
            json.dump(
                {
                    "round_id": report.round_id,
                    "start_time": report.start_time,
                    "end_time": report.end_time,
                    "num_clients": report.num_clients,
                    "agg_metrics": report.agg_metrics,
                    "status": report.status,
                    "client_metrics": report.client_metrics,
                    "phases_ms": job["phases_ms"],
                    "trainer_info": report.trainer_info,
                },
                f, indent=4,
            )
        os.replace(tmp, path)
        if self._recovery is not None:
            from nanofed_amd.server.fault_tolerance import RoundState

            version = self._model_manager.current_version
            self._recovery.checkpoint_round(
                report.round_id,
                {},  # client models stay on the device; the global state is enough to resume
                version.version_id if version is not None else "",
                {"model_state": state, "next_round": job["next_round"]},
                RoundState.COMPLETED,
            )

    def _retire_stale_checkpoints(self) -> None:
        """A fresh (not resumed) run must not recover into an older run's
        checkpoints that happen to live under the same directory: they are moved
        aside, once, before this run writes or reads its first checkpoint."""
        if self._stale_checked or self._recovery is None:
            return
        self._stale_checked = True
        if self._resumed:
            return
        ck = self.base_dir / "checkpoints"
        if ck.exists() and any(ck.glob("round_*")):
            stamp = get_current_time().strftime("%Y%m%d_%H%M%S")
            target = self.base_dir / f"checkpoints.stale-{stamp}"
            n = 0
            while target.exists():
                n += 1
                target = self.base_dir / f"checkpoints.stale-{stamp}-{n}"
            ck.rename(target)
            ck.mkdir()
            self.logger.warning(f"Moved checkpoints of an earlier run to {target.name}")

    def flush(self) -> None:
        """Wait until every submitted round is on disk (raises what the writer
        thread raised)."""
        if self._persister is not None:
            self._persister.flush()

    def close(self) -> None:
        if self._watchdog is not None:
            self._watchdog.close()
        if self._persister is not None:
            self._persister.close()
            self._persister = None

    def _on_round_timeout(self, round_id: int, limit_s: float) -> None:
        msg = f"Round {round_id} exceeded its time limit of {limit_s:g} s"
        try:
            self.logger.error(msg + f"; ending the process with status {WATCHDOG_EXIT_CODE}")
            if self.base_dir is not None and self.comm.is_root:
                (self.base_dir / "metrics").mkdir(parents=True, exist_ok=True)
                path = self.base_dir / "metrics" / f"metrics_round_{round_id}.json"
                with path.open("w") as f:
                    json.dump(
                        {
                            "round_id": round_id, "status": "FAILED", "num_clients": 0,
                            "agg_metrics": {}, "client_metrics": [],
                            "end_time": get_current_time().isoformat(),
                            "error": f"TimeoutError: {msg}",
                        },
                        f, indent=4,
                    )
        finally:
            os._exit(WATCHDOG_EXIT_CODE)

    # -- public helpers ---------------------------------------------------
    def run(self, num_rounds: int | None = None) -> list[RoundReport]:
        """Run rounds until ``num_rounds`` more have COMPLETED or FAILED on
        quorum.  A recoverable exception inside a round (``RuntimeError`` is how
        HIP and RCCL errors surface; reference: nanofed/server/fault_tolerance.py
        :139-152) rolls the global model back to the last completed checkpoint and
        repeats from there, at most ``max_recoveries`` times.  In a multi-rank job
        the exception every rank sees is ``RoundFailure``, raised after the round's
        collectives, so the recovery collectives match on all ranks."""
        out: list[RoundReport] = []
        target = num_rounds if num_rounds is not None else self.cfg.num_rounds
        self._in_run = True
        try:
            while len(out) < target:
                try:
                    self._more_rounds = len(out) + 1 < target
                    out.append(self.run_round())
                except Exception as failure:  # noqa: BLE001 - the strategy decides
                    cause = failure
                    if isinstance(failure, RoundFailure) and failure.local is not None:
                        cause = failure.local
                    if not self._recover(failure):
                        if cause is failure:
                            raise
                        raise cause from failure
                    # rounds after the recovery point are repeated: drop their reports
                    out = [rep for rep in out if rep.round_id < self.round_id]
        finally:
            self._in_run = False
        self.flush()
        return out

    def _recover(self, failure: Exception) -> bool:
        """Every rank takes the same decision: rank 0 owns the checkpoints, the
        restored arena and round counter are broadcast.  The trainers keep no
        state across rounds (step counters, noise offsets and the privacy spent
        are functions of the round number), so restoring the arena and the
        counter restores the run."""
        if not self.cfg.checkpoint_rounds or self.recoveries >= self.cfg.max_recoveries:
            return False
        self.recoveries += 1
        self._prelaunched = None
        self._timer.collect()
        self._tail_graph = None  # recaptured on a quiet round
        self._segment_graphs = {}
        self._tail_rounds = 0
        if self._persister is not None:
            try:
                self._persister.flush()
            except Exception:  # noqa: BLE001 - the failure being handled
                pass
        flag = torch.zeros(2, dtype=torch.float32, device=self.device)
        if self.comm.is_root and self._recovery is not None:
            self._retire_stale_checkpoints()
            # a failure that only a peer saw is recoverable by construction
            # (it reached us as RoundFailure, a RuntimeError)
            restored = self._recovery.handle_failure(failure, self.round_id)
            if restored is not None:
                _, state = restored
                self.global_arena.load_state_dict(state["model_state"])
                flag[0] = 1.0
                flag[1] = float(state.get("next_round", self.round_id))
            elif self._recovery.is_recoverable(failure):
                # nothing completed yet: the model the failed round started from
                self.global_arena.data.copy_(self._round_start)
                flag[0] = 1.0
                flag[1] = float(self.round_id)
        self.comm.broadcast(flag, src=0)
        if float(flag[0]) == 0.0:
            return False
        self.comm.broadcast(self.global_arena.data, src=0)
        self.round_id = int(flag[1])
        self.logger.warning(
            f"Recovered from {type(failure).__name__}: {failure}; resuming at round {self.round_id}"
        )
        return True

    def resume(self) -> int:
        """Continue a run from ``base_dir``: load the newest ModelManager version
        and take the round counter from its config (the reference always
        restarts at round 0, nanofed/server/model_manager/manager.py:144-188).
        Returns the round the engine will run next."""
        self._resumed = True
        self._prelaunched = None
        state = torch.zeros(1, dtype=torch.float32, device=self.device)
        if self.comm.is_root and self._model_manager is not None:
            versions = self._model_manager.list_versions()
            if versions:
                version = self._model_manager.load_model()
                self.global_arena.load_state_dict(self._host_model.state_dict())
                state[0] = float(version.config.get("round_id", -1) + 1)
        self.comm.broadcast(state, src=0)
        self.comm.broadcast(self.global_arena.data, src=0)
        self.round_id = int(state[0])
        return self.round_id

    def evaluate(self) -> dict[str, float]:
        return self.trainer.evaluate(self.global_arena.data)

    def global_state_dict(self) -> dict[str, torch.Tensor]:
        """CPU fp32 ``state_dict`` of the global model (checkpoint format)."""
        return self.global_arena.state_dict("cpu")

    def load_global_state_dict(self, state: dict) -> None:
        self.global_arena.load_state_dict(state)
        self._prelaunched = None  # a round queued early started from the old model

    def synchronize(self) -> None:
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def summary(self) -> dict:
        return {"config": asdict(self.cfg), "rounds": [asdict(r) for r in self.reports]}
