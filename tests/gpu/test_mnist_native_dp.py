"""Native per-sample DP-SGD (two-pass backward with ghost norms) against a
naive per-sample PyTorch oracle."""

import math

import pytest
import torch
import torch.nn.functional as F

from nanofed_amd.engine import DPConfig, EngineConfig, FederatedEngine
from nanofed_amd.models import MNISTModel
from nanofed_amd.parallel import Communicator

pytestmark = pytest.mark.gpu
B = 64


def make(noise=1e-6, clip=0.5, clients=2, **kw):
    torch.cuda.set_device(0)
    comm = Communicator(device=torch.device("cuda", 0))
    cfg = EngineConfig(model="mnist_cnn", num_clients=clients, samples_per_client=2 * B,
                       batch_size=B, local_epochs=1, eval_samples=128, graph_capture=False,
                       learning_rate=0.05,
                       dp=DPConfig(max_gradient_norm=clip, noise_multiplier=noise), **kw)
    return FederatedEngine(cfg, comm=comm)


def test_backend_is_native_and_clip_factors_match_a_per_sample_oracle():
    eng = make()
    assert type(eng.trainer).__name__ == "MnistCnnNativeTrainer"
    buf = eng.trainer.buf
    buf.ctx.p_drop1 = buf.ctx.p_drop2 = 0.0
    eng.pool.broadcast_from(eng.global_arena.data)
    # scale the weights up so that some samples are clipped and some are not
    eng.pool.data.mul_(1.0)
    eng.trainer._fill_indices(0, 0)
    buf.nvalid[1, 0] = 40
    buf.pack()
    before = [eng.pool[k].state_dict("cuda") for k in range(2)]
    buf.train_steps(0, 1)
    torch.cuda.synchronize()
    lr, C = eng.cfg.learning_rate, eng.cfg.dp.max_gradient_norm
    idx = buf.idx[:, :B].long()
    x = ((buf.images[idx].float() / 255.0 - buf.ctx.mean) * buf.ctx.inv_std).view(2, B, 1, 28, 28)
    y = buf.labels[idx].long()
    for k in range(2):
        nv = int(buf.nvalid[k, 0])
        model = MNISTModel().cuda().eval()
        model.load_state_dict(before[k])
        params = list(model.parameters())
        total = [torch.zeros_like(p) for p in params]
        factors = []
        for b in range(nv):
            grads = torch.autograd.grad(F.nll_loss(model(x[k, b : b + 1]), y[k, b : b + 1]), params)
            norm = math.sqrt(sum(float(g.pow(2).sum()) for g in grads))
            c = min(1.0, C / (norm + 1e-6))
            factors.append(c)
            for t, g in zip(total, grads):
                t += c * g / nv
        got_c = buf.dp_rowscale[k, :nv].cpu()
        want_c = torch.tensor(factors)
        assert 0.05 < float((want_c < 1).float().mean()), "test needs clipped samples"
        torch.testing.assert_close(got_c, want_c, rtol=5e-2, atol=1e-3)
        after = eng.pool[k].state_dict("cuda")
        for (name, p), t in zip(model.named_parameters(), total):
            step = (before[k][name] - after[name]) / lr
            cos = F.cosine_similarity(step.flatten(), t.flatten(), dim=0)
            assert float(cos) > 0.99, (name, float(cos))
            assert float(step.norm()) == pytest.approx(float(t.norm()), rel=8e-2)


def test_noise_has_the_dp_sgd_scale_and_changes_every_step():
    eng = make(noise=1.0, clip=1.0)
    buf = eng.trainer.buf
    eng.pool.broadcast_from(eng.global_arena.data)
    eng.trainer._fill_indices(0, 0)
    buf.pack()
    start = eng.pool.data.clone()
    buf.train_steps(0, 1)
    noisy = eng.pool.data.clone()
    # same step without noise, same dropout stream
    eng.pool.data.copy_(start)
    buf.step_counter.zero_()
    buf.ctx.dp_noise_std = 0.0
    buf.pack()
    buf.train_steps(0, 1)
    clean = eng.pool.data.clone()
    seg = eng.layout["fc1.weight"]
    diff = (noisy - clean)[:, seg.offset : seg.offset + seg.numel] / eng.cfg.learning_rate
    assert float(diff.std()) == pytest.approx(1.0 * 1.0 / B, rel=0.03)
    assert abs(float(diff.mean())) < 1e-4
    # clients draw independent noise
    assert float((diff[0] - diff[1]).abs().mean()) > 0.005


def test_dp_rounds_run_in_a_graph_report_epsilon_and_still_learn():
    torch.cuda.set_device(0)
    comm = Communicator(device=torch.device("cuda", 0))
    cfg = EngineConfig(model="mnist_cnn", num_clients=4, samples_per_client=1024, batch_size=B,
                       local_epochs=2, eval_samples=1000, learning_rate=0.5,
                       partition="dirichlet", dirichlet_alpha=0.5,
                       dp=DPConfig(max_gradient_norm=1.0, noise_multiplier=0.4))
    eng = FederatedEngine(cfg, comm=comm)
    reports = eng.run(6)
    assert all(r.status == "COMPLETED" for r in reports)
    eps = [r.agg_metrics["privacy_epsilon"] for r in reports]
    assert eps == sorted(eps) and eps[0] > 0
    assert torch.isfinite(eng.global_arena.data).all()
    assert eng.evaluate()["accuracy"] > 0.3  # chance is 0.1; clipped, noisy, non-IID


@pytest.mark.parametrize("capture", [True, False])
def test_noisy_rounds_are_reproducible_bit_for_bit(capture):
    """The noise of a step is keyed by (client seed, the client's step counter).  The fused
    last launch of the step both draws noise by that counter in all of its workgroups and
    used to advance the counter from one of them: workgroups that ran after it drew the next
    step's noise - a different model from run to run, and for those elements the same noise
    in two consecutive steps.  The counter now advances in a launch of its own behind it:
    two engines with the same seed end with the same bits, and the counters count the steps."""
    outs = []
    for _ in range(2):
        torch.cuda.set_device(0)
        comm = Communicator(device=torch.device("cuda", 0))
        cfg = EngineConfig(model="mnist_cnn", num_clients=4, samples_per_client=1024,
                           batch_size=B, local_epochs=2, eval_samples=256, learning_rate=0.1,
                           graph_capture=capture,
                           dp=DPConfig(max_gradient_norm=1.0, noise_multiplier=1.1))
        eng = FederatedEngine(cfg, comm=comm)
        reports = eng.run(2)
        torch.cuda.synchronize()
        assert all(r.status == "COMPLETED" for r in reports)
        steps = eng.trainer.buf.step_counter.clone() if hasattr(eng.trainer, "buf") else None
        outs.append((eng.global_arena.data.clone(), [r.agg_metrics["loss"] for r in reports], steps))
        eng.close()
    assert outs[0][1] == outs[1][1]
    assert torch.equal(outs[0][0], outs[1][0])
    if outs[0][2] is not None:
        # 2 rounds x 2 epochs x 16 steps per client
        assert outs[0][2].tolist() == [64] * 4
