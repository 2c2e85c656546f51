"""The host-API server classes (aggregators, validators, in-node transport) fed
with device-resident arenas: they must run the fused kernels and agree with the
CPU results."""

import asyncio

import pytest
import torch
import torch.nn as nn

from nanofed_amd import Coordinator, CoordinatorConfig, FedAvgAggregator, ModelManager, coordinate
from nanofed_amd.communication import InNodeClient, InNodeServer
from nanofed_amd.core import ModelUpdate
from nanofed_amd.models import MNISTModel
from nanofed_amd.runtime import ArenaLayout, ParameterArena
from nanofed_amd.server.aggregator import PrivacyAwareAggregationConfig, PrivacyAwareAggregator
from nanofed_amd.server.aggregator.secure import PairwiseMaskSecureAggregation
from nanofed_amd.server.validation import DefaultModelValidator, ValidationConfig, ValidationResult
from nanofed_amd.utils import get_current_time

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _update(i, state, n, rnd=0):
    return ModelUpdate(client_id=f"c{i}", round_number=rnd, model_state=state,
                       metrics={"samples_processed": n, "loss": 0.1 * i}, timestamp=get_current_time())


def test_fedavg_on_device_arenas_equals_cpu():
    torch.manual_seed(0)
    models = [MNISTModel() for _ in range(4)]
    layout = ArenaLayout.from_module(models[0])
    counts = [100, 300, 200, 400]
    arenas = []
    for m in models:
        a = ParameterArena(layout, DEV)
        a.load_state_dict(m.state_dict())
        arenas.append(a)
    cpu_model, gpu_model = MNISTModel(), MNISTModel().to(DEV)
    FedAvgAggregator().aggregate(cpu_model, [_update(i, m.state_dict(), n) for i, (m, n) in enumerate(zip(models, counts))])
    res = FedAvgAggregator().aggregate(gpu_model, [_update(i, a.views(), n) for i, (a, n) in enumerate(zip(arenas, counts))])
    assert res.num_clients == 4
    for (k, p), q in zip(cpu_model.state_dict().items(), gpu_model.state_dict().values()):
        assert q.device.type == "cuda"
        torch.testing.assert_close(q.cpu(), p, rtol=1e-5, atol=1e-6, msg=k)


def test_central_privacy_aggregation_on_device_clips_and_reports_budget():
    torch.manual_seed(1)
    net = nn.Sequential(nn.Linear(32, 16), nn.ReLU(), nn.Linear(16, 4))
    layout = ArenaLayout.from_module(net)
    ups = []
    for i in range(3):
        a = ParameterArena(layout, DEV)
        a.load_state_dict({k: v * (5.0 + i) for k, v in net.state_dict().items()})
        ups.append(_update(i, a.views(), 100))
    cfg = PrivacyAwareAggregationConfig(epsilon=8.0, delta=1e-5, max_gradient_norm=1.0,
                                        noise_multiplier=0.01, min_clients=2)
    out = nn.Sequential(nn.Linear(32, 16), nn.ReLU(), nn.Linear(16, 4)).to(DEV)
    res = PrivacyAwareAggregator(cfg).aggregate(out, ups)
    flat = torch.cat([p.detach().reshape(-1) for p in out.state_dict().values()])
    # every client state was clipped to norm 1 before averaging (noise is tiny)
    assert float(flat.norm()) <= 1.05
    assert res.metrics["privacy_epsilon"] > 0 and res.metrics["privacy_delta"] == pytest.approx(1e-5)


def test_pairwise_masks_cancel_on_device():
    n = 4
    proto = PairwiseMaskSecureAggregation(num_clients=n, mask_scale=4.0, secret=b"k" * 32)
    flats = [torch.randn(1 << 16, device=DEV) for _ in range(n)]
    masked = [proto.mask(i, f) for i, f in enumerate(flats)]
    assert masked[0].device.type == "cuda"
    assert float((masked[0] - flats[0]).abs().mean()) > 0.5
    torch.testing.assert_close(proto.aggregate_flat(masked), sum(flats), atol=2e-3, rtol=0)
    present = [0, 1, 3]
    torch.testing.assert_close(proto.aggregate_flat([masked[i] for i in present], present=present),
                               sum(flats[i] for i in present), atol=2e-3, rtol=0)


def test_validator_verdicts_on_device_arena():
    net = MNISTModel()
    layout = ArenaLayout.from_module(net)
    good = ParameterArena(layout, DEV)
    good.load_state_dict(net.state_dict())
    bad = ParameterArena(layout, DEV)
    bad.load_state_dict(net.state_dict())
    bad.data[12345] = float("nan")
    huge = ParameterArena(layout, DEV)
    huge.load_state_dict({k: v * 1000 for k, v in net.state_dict().items()})
    vcfg = ValidationConfig(max_norm=50.0)
    v = DefaultModelValidator(vcfg)
    shapes = {k: tuple(t.shape) for k, t in net.state_dict().items()}
    assert v.validate_shape(_update(0, good.views(), 1), shapes) == ValidationResult.VALID
    assert v.validate_range(_update(0, good.views(), 1), vcfg) == ValidationResult.VALID
    assert v.validate_range(_update(1, bad.views(), 1), vcfg) == ValidationResult.INVALID_RANGE
    assert v.validate_range(_update(2, huge.views(), 1), vcfg) == ValidationResult.INVALID_RANGE


def test_coordinator_over_in_node_transport_keeps_everything_on_the_gpu(tmp_path):
    async def client_loop(cid, server, samples, shift):
        async with InNodeClient(server.url, cid) as client:
            done = set()
            while not await client.check_server_status():
                state, rnd = await client.fetch_global_model()
                if rnd in done:
                    await asyncio.sleep(0.01)
                    continue
                assert all(t.device.type == "cuda" for t in state.values())
                model = MNISTModel().to(DEV)
                model.load_state_dict(state)
                with torch.no_grad():
                    for p in model.parameters():
                        p.add_(shift)
                arena = ParameterArena.from_module(model)
                assert arena.device.type == "cuda"
                if await client.submit_update(arena.views(), {"loss": 1.0, "samples_processed": samples}):
                    done.add(rnd)

    async def go():
        server = InNodeServer("gpu")
        await server.start()
        torch.manual_seed(0)
        manager = ModelManager(MNISTModel().to(DEV))
        start = {k: v.clone() for k, v in manager.model.state_dict().items()}
        coord = Coordinator(manager, FedAvgAggregator(), server, CoordinatorConfig(2, 2, 1.0, 60, tmp_path))
        await asyncio.gather(coordinate(coord), client_loop("a", server, 300, 1.0),
                             client_loop("b", server, 100, -1.0))
        await server.stop()
        return start, manager

    start, manager = asyncio.run(go())
    for k, v in manager.model.state_dict().items():
        if v.dtype.is_floating_point:
            torch.testing.assert_close(v, start[k] + 1.0, rtol=1e-5, atol=1e-5)
    # the checkpoints are CPU fp32 state dicts in the reference's format
    files = sorted((tmp_path / "models" / "models").glob("*.pt"))
    state = torch.load(files[-1], weights_only=True)
    assert all(t.device.type == "cpu" for t in state.values()) and "conv1.weight" in state
