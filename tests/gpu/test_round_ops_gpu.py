"""HIP kernels of the round tail (csrc/hip/arena_ops.hip) against the fp32 PyTorch
implementations of the same operations, and the captured tail graph of the engine
against eager execution over several replays."""

import pytest
import torch

from nanofed_amd import ops
from nanofed_amd.engine import EngineConfig, FaultPlan, FederatedEngine
from nanofed_amd.parallel import Communicator

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("G,n", [(1, 37), (8, 1199882), (5, 65537)])
def test_screen_rows_matches_the_cpu_oracle(G, n):
    torch.manual_seed(G * 1000 + n % 97)
    pitch = (n + 3) // 4 * 4  # rows of a pool start on 16-byte boundaries
    rows = torch.randn(G, pitch)[:, :n]
    if G > 2:
        rows[2, n // 2] = float("inf")
        rows[1] *= 40.0
    ref = torch.randn(n)
    rec = torch.zeros(G + 1, 8)
    rec[:G, 3] = 1.0
    want = rec.clone()
    bound = 20.0 * (n ** 0.5)
    ops.screen_rows(rows, want[:G], col_sq=4, col_valid=3, ref=ref, max_norm=bound)
    dev_rows = torch.zeros(G, pitch, device=DEV)
    dev_rows[:, :n] = rows.to(DEV)
    dev_rows = dev_rows[:, :n]
    got = rec.to(DEV)
    ops.screen_rows(dev_rows, got[:G], col_sq=4, col_valid=3, ref=ref.to(DEV), max_norm=bound)
    torch.testing.assert_close(got.cpu()[:, 4], want[:, 4], rtol=2e-5, atol=1e-3)
    assert got.cpu()[:, 3].tolist() == want[:, 3].tolist()
    assert got.cpu()[G].abs().sum() == 0  # the row after the last one is not touched
    # without a reference: finite check of the rows themselves
    got2 = rec.to(DEV)
    ops.screen_rows(dev_rows, got2[:G], col_sq=4, col_valid=3)
    want2 = rec.clone()
    ops.screen_rows(rows, want2[:G], col_sq=4, col_valid=3)
    assert got2.cpu()[:, 3].tolist() == want2[:, 3].tolist()


def test_screen_rows_refuses_rows_that_are_not_16_byte_aligned():
    from nanofed_amd.core.exceptions import DeviceError

    rows = torch.randn(3, 1001, device=DEV)
    with pytest.raises(DeviceError):
        ops.screen_rows(rows, torch.zeros(3, 8, device=DEV), 4, 3)


def test_screen_rows_is_run_to_run_identical():
    torch.manual_seed(3)
    rows = torch.randn(8, 300004, device=DEV)[:, :300001]
    ref = torch.randn(300001, device=DEV)
    outs = []
    for _ in range(3):
        rec = torch.zeros(8, 8, device=DEV)
        rec[:, 3] = 1.0
        ops.screen_rows(rows, rec, 4, 3, ref=ref)
        outs.append(rec.cpu())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("z,stats", [(0.0, 5), (2.0, 5), (2.0, 12)])
def test_round_weights_matches_the_cpu_oracle(z, stats):
    torch.manual_seed(5)
    K = 9
    rec = torch.zeros(K, 8)
    rec[:, 0] = torch.randint(100, 5000, (K,)).float()
    rec[:, 3] = torch.tensor([1, 1, 0, 1, 1, 1, 1, 1, 1.0])
    norms = 1.0 + 0.1 * torch.randn(K)
    norms[6] = 60.0
    rec[:, 4] = norms.square()
    for required in (3.0, 9.0):
        want_rec, got_rec = rec.clone(), rec.clone().to(DEV)
        w, f = torch.zeros(K), torch.zeros(4)
        gw, gf = torch.zeros(K, device=DEV), torch.zeros(4, device=DEV)
        ops.round_weights(want_rec, 0, 3, 4, required, w, f, z_threshold=z, min_clients_for_stats=stats)
        ops.round_weights(got_rec, 0, 3, 4, required, gw, gf, z_threshold=z, min_clients_for_stats=stats)
        torch.testing.assert_close(gw.cpu(), w, rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(gf.cpu(), f, rtol=1e-6, atol=0)
        assert got_rec.cpu()[:, 3].tolist() == want_rec[:, 3].tolist()
        if z > 0 and stats <= 8:
            assert want_rec[6, 3] == 0.0 and f[3] == 1.0


def test_select_copy_on_the_device():
    dst = torch.zeros(100003, device=DEV)
    src = torch.randn(100003, device=DEV)
    ops.select_copy(dst, src, torch.zeros(4, device=DEV))
    assert float(dst.abs().sum()) == 0.0
    ops.select_copy(dst, src, torch.ones(4, device=DEV))
    assert torch.equal(dst, src)


def test_pairwise_masks_match_the_cpu_streams_and_cancel():
    torch.manual_seed(1)
    K, n, scale = 6, 100003, 1.5
    seeds = torch.zeros(K, K, dtype=torch.int64)
    for i in range(K):
        for j in range(i + 1, K):
            seeds[i, j] = seeds[j, i] = (1 << 40) + 1000 * i + j
    rows = [torch.randn(n) for _ in range(K)]
    masked = []
    for i in range(K):
        peers = [j for j in range(K) if j != i]
        args = ([int(seeds[i, j]) for j in peers], [1.0 if i < j else -1.0 for j in peers], scale)
        cpu = rows[i].clone()
        ops.pairwise_mask_(cpu, *args, prescale=0.25, offset=11)
        gpu = rows[i].to(DEV)
        ops.pairwise_mask_(gpu, *args, prescale=0.25, offset=11)
        torch.testing.assert_close(gpu.cpu(), cpu, rtol=0, atol=2e-6)
        masked.append(gpu)
    total = torch.stack(masked).sum(0)
    torch.testing.assert_close(total.cpu(), 0.25 * torch.stack(rows).sum(0), rtol=0, atol=3e-5)
    present = torch.tensor([1.0, 1.0, 0.0, 1.0, 0.0, 1.0])
    part = sum(masked[i] for i in (0, 1, 3, 5))
    ops.pairwise_repair_(part, present.to(DEV), seeds.to(DEV), scale, offset=11)
    torch.testing.assert_close(part.cpu(), 0.25 * (rows[0] + rows[1] + rows[3] + rows[5]),
                               rtol=0, atol=3e-5)


CFG = dict(model="mnist_cnn", num_clients=8, samples_per_client=256, local_epochs=1,
           eval_samples=128, max_update_norm=1e6, zscore_threshold=3.0)


def test_tail_graph_replays_equal_eager_tails_round_after_round():
    """Six rounds with the captured tail against six with the eager tail.  Between
    the replays the program does its own eager work (training graphs, the host
    read of the records, an evaluation), as a real run does."""
    graph = FederatedEngine(EngineConfig(tail_graph=True, **CFG), comm=Communicator(device=torch.device("cuda", 0)))
    eager = FederatedEngine(EngineConfig(tail_graph=False, **CFG), comm=Communicator(device=torch.device("cuda", 0)))
    for r in range(6):
        a, b = graph.run_round(), eager.run_round()
        assert torch.equal(graph.global_arena.data, eager.global_arena.data), f"round {r}"
        assert a.agg_metrics == b.agg_metrics and a.status == b.status == "COMPLETED"
        assert [c["weight"] for c in a.client_metrics] == [c["weight"] for c in b.client_metrics]
        assert [c["update_norm"] for c in a.client_metrics] == [c["update_norm"] for c in b.client_metrics]
        if r % 2:
            assert graph.evaluate() == eager.evaluate()
            torch.cuda.synchronize()
    assert graph._tail_graph is not None and eager._tail_graph is None
    graph.close()
    eager.close()


def test_a_faulty_round_between_replays_runs_eagerly_and_the_graph_stays_valid():
    faults = dict(drop={3: {2}}, poison={4: {5}})
    graph = FederatedEngine(EngineConfig(tail_graph=True, **CFG), comm=Communicator(device=torch.device("cuda", 0)),
                            faults=FaultPlan(**faults))
    eager = FederatedEngine(EngineConfig(tail_graph=False, **CFG), comm=Communicator(device=torch.device("cuda", 0)),
                            faults=FaultPlan(**faults))
    for r in range(7):
        a, b = graph.run_round(), eager.run_round()
        assert torch.equal(graph.global_arena.data, eager.global_arena.data), f"round {r}"
        assert a.num_clients == b.num_clients
        assert [c["valid"] for c in a.client_metrics] == [c["valid"] for c in b.client_metrics]
        if r == 3:
            assert not a.client_metrics[2]["valid"]
        if r == 4:
            assert not a.client_metrics[5]["valid"]
    graph.close()
    eager.close()


def _engine(faults=None, **kw):
    cfg = dict(model="mnist_cnn", num_clients=8, samples_per_client=256, local_epochs=1,
               eval_samples=128)
    cfg.update(kw)
    return FederatedEngine(EngineConfig(**cfg), comm=Communicator(device=torch.device("cuda", 0)),
                           faults=FaultPlan(**(faults or {})))


def test_a_fifty_fold_update_is_masked_out_by_the_z_score_screen():
    eng = _engine(faults=dict(scale={0: {5: 50.0}}), zscore_threshold=2.0)
    ref = _engine(faults=dict(drop={0: {5}}))
    a, b = eng.run_round(), ref.run_round()
    assert [c["valid"] for c in a.client_metrics] == [True] * 5 + [False] + [True] * 2
    assert a.agg_metrics["anomalous_clients"] == 1.0 and a.num_clients == 7
    norms = [c["update_norm"] for c in a.client_metrics]
    assert norms[5] > 20 * max(norms[:5] + norms[6:])
    # leaving the outlier out is the same as the client never reporting
    torch.testing.assert_close(eng.global_arena.data, ref.global_arena.data, rtol=0, atol=0)
    assert b.num_clients == 7
    eng.close()
    ref.close()


@pytest.mark.parametrize("drop", [None, {0: {2, 6}}])
def test_a_pairwise_masked_round_equals_the_plain_round(drop):
    faults = dict(drop=drop) if drop else {}
    plain = _engine(faults=dict(faults))
    blind = _engine(faults=dict(faults), secure_agg="pairwise")
    a, b = plain.run_round(), blind.run_round()
    assert a.num_clients == b.num_clients == (6 if drop else 8)
    torch.testing.assert_close(blind.global_arena.data, plain.global_arena.data, rtol=0, atol=1e-5)
    # what crosses the reduce is blinded: a masked row is far from the client's model
    row = blind.pool[0].data
    assert float((row - plain.pool[0].data * blind._shares_host[0]).abs().mean()) > 0.1
    plain.close()
    blind.close()
