"""Device-plane configuration.

The reference configures a run through constructor dataclasses
(``TrainingConfig`` nanofed/trainer/base.py:15-24, ``CoordinatorConfig``
nanofed/orchestration/coordinator.py:26-49) and has no notion of devices.
Those classes are kept as they are; the knobs that only exist on the device
plane (clients per GPU, compute dtype, graph capture, reduce algorithm, data
sharding, DP-SGD) live here.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Sequence


@dataclass(frozen=True)
class DPConfig:
    """Per-step DP-SGD on the device (clip + Gaussian noise on gradients).

    ``mode='per_sample'`` clips every example's gradient (Abadi et al.);
    ``mode='batch'`` clips the batch-mean gradient, which is what the
    reference's ``PrivateTrainer`` does (nanofed/trainer/private.py:54-63).
    """

    max_gradient_norm: float = 1.0
    noise_multiplier: float = 1.1
    epsilon: float = 10.0
    delta: float = 1e-5
    mode: str = "per_sample"


@dataclass(frozen=True)
class EngineConfig:
    model: str = "mnist_cnn"
    num_clients: int = 8
    num_rounds: int = 2
    local_epochs: int = 2
    batch_size: int = 64
    learning_rate: float = 0.1
    weight_decay: float = 0.0
    # data
    dataset: str = "synthetic_mnist"  # synthetic_mnist[_hard] | mnist | synthetic_cifar10[_hard] | cifar10 | synthetic_tokens | tokens
    samples_per_client: int | Sequence[int] = 7500
    partition: str = "iid"  # iid | dirichlet
    dirichlet_alpha: float = 0.1
    eval_samples: int = 10000
    data_dir: str = "./data"
    seq_len: int = 1024
    # synthetic token streams: confine the Markov chain to the first ids of the
    # vocabulary (0 = all of it); the model keeps its full width
    token_active_vocab: int = 0
    # dataset "tokens": element type of <data_dir>/train.bin and val.bin (uint16 | uint32 | ...)
    token_dtype: str = "uint16"
    # execution
    backend: str = "auto"  # auto | native | torch
    dtype: str = "bf16"  # compute dtype of the local trainers: bf16 | fp32
    graph_capture: bool = True
    # torch trainer: train the clients of a GPU concurrently, one HIP stream each
    # (their kernels are too small to fill the chip one client at a time)
    client_streams: bool = True
    # how many streams the clients of a GPU share (a process has 4 hardware queues
    # by default; more streams than queues gain nothing)
    client_stream_count: int = 4
    # native GPT-2 trainer: concurrent client steps ("lanes", at most 2; see the
    # note in GPT2NativeTrainer.__init__ before raising this from 1)
    gpt2_lanes: int = 1
    # FedAvg reduce over the ranks: "ordered" adds the client rows in client order on
    # every rank's slice (all-to-all of row slices + all-gather; the global model is then
    # bit-identical for any number of ranks), "allreduce" / "rs_ag" add per-rank partial
    # sums with the library's all-reduce / reduce-scatter + all-gather (order up to the
    # library).  "auto": ordered for arenas below 32 MB (where either is latency-bound),
    # rs_ag above.
    reduce_algo: str = "auto"  # auto | ordered | allreduce | rs_ag
    # True: local training does not depend on how many clients share a GPU either (the
    # native MNIST kernels keep ONE workgroup geometry instead of choosing it from the
    # number of clients on the GPU), so that together with the ordered reduce a run gives
    # the same bits on 1, 2, 4 or 8 GPUs - the mode in which a multi-GPU run can be
    # checked against a single-GPU one.  Costs speed where few clients share a GPU.
    placement_invariant: bool = False
    drop_last: bool = False
    # native MNIST kernels: conv2 workgroups per image (1 or 3); 0 = choose from
    # the number of clients on the GPU
    conv_split: int = 0
    # quorum / robustness (reference: CoordinatorConfig.min_completion_rate)
    min_clients: int = 1
    min_completion_rate: float = 1.0
    validate_updates: bool = True
    max_update_norm: float = 0.0  # 0 disables the norm bound (finite check stays)
    # anomaly screen on the update norms ||theta_k - theta|| (reference:
    # ValidationConfig.z_score_threshold / min_clients_for_stats,
    # nanofed/server/validation.py:25-33,103-135): a client further than this many
    # standard deviations from the mean of the OTHER valid clients is left out;
    # 0 disables the screen
    zscore_threshold: float = 0.0
    min_clients_for_stats: int = 5
    # secure aggregation on the device plane: "pairwise" blinds every client row
    # with seeded pairwise masks that cancel in the reduce (dropped clients'
    # unmatched masks are removed on device); "none" reduces the plain rows
    secure_agg: str = "none"
    secure_mask_scale: float = 1.0
    # Mixed rounds: this many extra arena rows take updates that clients outside the
    # engine submitted for the round (FederatedEngine.submit_external; DeviceCoordinator
    # fills them from its server) - same screening, same weighted sum, weight n_k.
    external_slots: int = 0
    # Pair seeds come from X25519 agreement between the clients' key pairs.  None: fresh
    # random keys per engine.  An integer derives the private keys from it - reproducible
    # masks for tests and debugging, and no secrecy (who knows the number knows the keys).
    secure_key_seed: int | None = None
    # signed updates (reference: SecurityManager.sign_update / verify_signature,
    # nanofed/server/validation.py:138-212): every client holds an RSA-2048 key and signs
    # (round, client id, tree digest of its arena row) with RSA-PSS; before aggregation the
    # digest of the row as it arrived is recomputed and the signature checked - a row that
    # fails is left out of the round.  Digests are computed on the device
    # (ops/digest_ops.py); costs two small host synchronisations per round.
    sign_updates: bool = False
    # a client that reports later than this many seconds is left out of the round
    # (reference: CoordinatorConfig.round_timeout); 0 = wait for everybody
    round_timeout_s: float = 0.0
    # True: a watchdog thread enforces round_timeout_s on the wall clock - a round
    # that overruns it (a stalled rank blocks every collective) is recorded as
    # FAILED and the process ends with exit code WATCHDOG_EXIT_CODE, so that a
    # supervisor can start a fresh process that resume()s (reference: TimeoutError
    # + RoundStatus.FAILED, nanofed/orchestration/coordinator.py:205-245,295-304).
    # The first round of a process gets first_round_grace_s on top (graph capture,
    # library warm-up).
    enforce_round_timeout: bool = False
    first_round_grace_s: float = 120.0
    # fault tolerance: write a FileStateStore checkpoint after every completed
    # round and roll back to the last one on a recoverable failure
    checkpoint_rounds: bool = False
    max_recoveries: int = 2
    # persistence (model version, metrics JSON, checkpoint) runs on a writer thread
    # fed by a device->pinned-host copy on a side stream; False writes inline
    async_persist: bool = True
    # capture the round's tail (screening, weights, weighted sum, select) in a
    # hipGraph (single-rank GPU runs)
    tail_graph: bool = True
    # in run(): queue the next round's broadcast and local training on the device before
    # the host waits for the current round's outcome (no device idle time between rounds)
    prefetch_rounds: bool = True
    # observability: HIP-event timing of the round's phases (extra key
    # "phases_ms" in the round metrics JSON)
    phase_timing: bool = True
    # privacy
    dp: DPConfig | None = None
    seed: int = 0
    # model kwargs (e.g. GPT-2 sizes for tests)
    model_kwargs: dict = field(default_factory=dict)

    def required_clients(self) -> int:
        """Updates needed to close a round: floor(min_clients * rate), never
        less than one (the reference's int() can reach zero,
        nanofed/orchestration/coordinator.py:209-211)."""
        return max(1, int(self.min_clients * self.min_completion_rate))
