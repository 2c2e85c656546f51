"""Per-sample DP step of the PyTorch group trainer against a naive loop of
per-example autograd (DP-SGD: clip every example, average, add noise)."""

import pytest
import torch
import torch.nn.functional as F

from nanofed_amd.core.exceptions import DeviceError
from nanofed_amd.engine import DPConfig, EngineConfig, FederatedEngine
from nanofed_amd.models import build_model
from nanofed_amd.parallel import Communicator

CFG = dict(model="mnist_mlp", num_clients=2, samples_per_client=32, batch_size=32, local_epochs=1,
           eval_samples=16, dtype="fp32", backend="torch", model_kwargs={"hidden": 8},
           learning_rate=0.5)


def test_one_noise_free_step_equals_the_per_example_oracle():
    clip = 0.05  # small enough that most examples are clipped
    cfg = EngineConfig(dp=DPConfig(max_gradient_norm=clip, noise_multiplier=0.0), **CFG)
    eng = FederatedEngine(cfg, comm=Communicator())
    before = eng.global_arena.state_dict("cpu")
    eng.run_round()
    for k in range(2):
        rows = torch.from_numpy(eng.data.shards[k])
        x = (eng.data.inputs[rows].float() / 255.0 - eng.shards.mean) / eng.shards.std
        y = eng.data.labels[rows]
        model = build_model("mnist_mlp", hidden=8)
        model.load_state_dict(before)
        total = {n: torch.zeros_like(p) for n, p in model.named_parameters()}
        clipped = 0
        for i in range(len(x)):
            model.zero_grad()
            F.nll_loss(model(x[i : i + 1]), y[i : i + 1]).backward()
            norm = torch.sqrt(sum(p.grad.square().sum() for p in model.parameters()))
            f = min(1.0, clip / float(norm))
            clipped += f < 1.0
            for n, p in model.named_parameters():
                total[n] += p.grad * f
        assert clipped > len(x) // 2
        after = eng.pool[k].state_dict("cpu")
        for n, p in model.named_parameters():
            want = before[n] - 0.5 * total[n] / len(x)
            torch.testing.assert_close(after[n], want, rtol=1e-4, atol=1e-6)
    eng.close()


def test_noise_has_the_dp_sgd_scale_and_epsilon_is_reported():
    sigma, clip = 2.0, 1e-3  # gradients vanish next to the noise
    cfg = EngineConfig(dp=DPConfig(max_gradient_norm=clip, noise_multiplier=sigma), **CFG)
    eng = FederatedEngine(cfg, comm=Communicator())
    before = eng.global_arena.data.clone()
    report = eng.run_round()
    n = eng.layout.trainable_numel
    step = (eng.pool[0].data[:n] - before[:n]) / 0.5
    assert float(step.std()) == pytest.approx(sigma * clip / 32, rel=0.15)
    assert report.agg_metrics["privacy_epsilon"] > 0
    second = eng.run_round()
    assert second.agg_metrics["privacy_epsilon"] > report.agg_metrics["privacy_epsilon"]
    eng.close()


def test_batch_mode_keeps_the_reference_step_and_per_sample_refuses_batchnorm():
    cfg = EngineConfig(dp=DPConfig(max_gradient_norm=1.0, noise_multiplier=0.0, mode="batch"), **CFG)
    eng = FederatedEngine(cfg, comm=Communicator())
    assert eng.run_round().status == "COMPLETED"
    eng.close()
    with pytest.raises(DeviceError, match="BatchNorm"):
        FederatedEngine(
            EngineConfig(model="resnet18", dataset="synthetic_cifar10", num_clients=1,
                         samples_per_client=8, batch_size=8, eval_samples=8, dtype="fp32",
                         backend="torch", dp=DPConfig()),
            comm=Communicator(),
        )
