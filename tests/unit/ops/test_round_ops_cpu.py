"""CPU semantics of the round-tail operations (the fp32 PyTorch implementations
that also serve as oracles of the HIP kernels): row screening, device-side FedAvg
weights with the z-score anomaly screen, flag-gated copy, pairwise blinding."""

import pytest
import torch

from nanofed_amd import ops


def test_screen_rows_writes_norms_and_verdicts():
    torch.manual_seed(0)
    rows = torch.randn(4, 1000)
    rows[2, 17] = float("nan")
    rows[3] *= 100.0
    ref = torch.randn(1000)
    rec = torch.zeros(4, 6)
    rec[:, 3] = 1.0
    ops.screen_rows(rows, rec, col_sq=4, col_valid=3, ref=ref, max_norm=200.0)
    delta = rows - ref
    want = torch.nan_to_num(delta, nan=0.0).square().sum(dim=1)
    torch.testing.assert_close(rec[:, 4], want)
    assert rec[:, 3].tolist() == [1.0, 1.0, 0.0, 0.0]  # NaN row and the over-norm row
    # a client that was already invalid stays invalid
    rec2 = torch.zeros(4, 6)
    ops.screen_rows(rows[:2], rec2, 4, 3)
    assert rec2[:2, 3].tolist() == [0.0, 0.0]


def test_round_weights_are_sample_proportions_of_valid_clients():
    rec = torch.zeros(4, 6)
    rec[:, 0] = torch.tensor([1000.0, 2000.0, 500.0, 500.0])
    rec[:, 3] = torch.tensor([1.0, 1.0, 0.0, 1.0])
    w, flags = torch.zeros(4), torch.zeros(4)
    ops.round_weights(rec, 0, 3, 4, required=2.0, weights=w, flags=flags)
    assert w.tolist() == pytest.approx([1000 / 3500, 2000 / 3500, 0.0, 500 / 3500])
    assert flags.tolist() == pytest.approx([1.0, 3.0, 3500.0, 0.0])
    ops.round_weights(rec, 0, 3, 4, required=4.0, weights=w, flags=flags)
    assert flags[0] == 0.0  # quorum missed


def test_z_score_screen_drops_the_outlier_against_the_other_clients():
    # reference test case: a x50 update among ordinary ones is ANOMALOUS
    # (nanofed/server/validation.py:103-135)
    rec = torch.zeros(8, 6)
    rec[:, 0] = 100.0
    rec[:, 3] = 1.0
    norms = torch.tensor([1.0, 1.1, 0.9, 1.05, 0.95, 1.0, 50.0, 1.02])
    rec[:, 4] = norms.square()
    w, flags = torch.zeros(8), torch.zeros(4)
    ops.round_weights(rec, 0, 3, 4, 1.0, w, flags, z_threshold=2.0, min_clients_for_stats=5)
    assert rec[:, 3].tolist() == [1, 1, 1, 1, 1, 1, 0, 1]
    assert w[6] == 0.0 and w.sum().item() == pytest.approx(1.0)
    assert flags[3] == 1.0
    # too few clients for statistics: nobody is dropped
    rec3 = torch.zeros(3, 6)
    rec3[:, 0], rec3[:, 3] = 10.0, 1.0
    rec3[:, 4] = torch.tensor([1.0, 1.0, 2500.0])
    w3 = torch.zeros(3)
    ops.round_weights(rec3, 0, 3, 4, 1.0, w3, flags, z_threshold=2.0, min_clients_for_stats=5)
    assert rec3[:, 3].tolist() == [1, 1, 1]


def test_select_copy_obeys_the_flag():
    dst, src = torch.zeros(10), torch.ones(10)
    ops.select_copy(dst, src, torch.tensor([0.0]))
    assert dst.sum() == 0
    ops.select_copy(dst, src, torch.tensor([1.0]))
    assert dst.sum() == 10


def test_pairwise_masks_cancel_and_repair_removes_unmatched_ones():
    torch.manual_seed(1)
    K, n, scale = 5, 4001, 2.0
    seeds = torch.zeros(K, K, dtype=torch.int64)
    for i in range(K):
        for j in range(i + 1, K):
            seeds[i, j] = seeds[j, i] = 1000 * i + j + 7
    rows = [torch.randn(n) for _ in range(K)]
    masked = []
    for i in range(K):
        peers = [j for j in range(K) if j != i]
        m = rows[i].clone()
        ops.pairwise_mask_(m, [int(seeds[i, j]) for j in peers],
                           [1.0 if i < j else -1.0 for j in peers], scale, prescale=0.5, offset=3)
        # a blinded row is far from the row it hides
        assert (m - 0.5 * rows[i]).abs().mean() > 0.5
        masked.append(m)
    total = torch.stack(masked).sum(0)
    torch.testing.assert_close(total, 0.5 * torch.stack(rows).sum(0), rtol=0, atol=2e-5)
    # clients 1 and 4 never arrive: their partners' masks are removed
    present = torch.tensor([1.0, 0.0, 1.0, 1.0, 0.0])
    part = sum(masked[i] for i in (0, 2, 3))
    ops.pairwise_repair_(part, present, seeds, scale, offset=3)
    torch.testing.assert_close(part, 0.5 * (rows[0] + rows[2] + rows[3]), rtol=0, atol=2e-5)
    # one mask of pairwise_mask_ equals mask_add_ with the same seed
    a, b = torch.zeros(64), torch.zeros(64)
    ops.pairwise_mask_(a, [99], [1.0], scale)
    ops.mask_add_(b, seed=99, scale=scale)
    assert torch.equal(a, b)
