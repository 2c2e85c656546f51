"""Update validation / signing, checkpoints and recovery, model versions
(nanofed/server/validation.py, fault_tolerance.py, model_manager/manager.py)."""

import json

import pytest
import torch
import torch.nn as nn

from nanofed_amd.core import ModelManagerError, ModelUpdate
from nanofed_amd.server import (
    CheckpointMetadata,
    FaultTolerantCoordinator,
    FileStateStore,
    ModelManager,
    RoundState,
    SimpleRecoveryStrategy,
)
from nanofed_amd.server.validation import (
    DefaultModelValidator,
    SecurityManager,
    ValidationConfig,
    ValidationResult,
    validation_mask,
)
from nanofed_amd.utils import get_current_time


def upd(state, cid="c", rnd=0):
    return ModelUpdate(client_id=cid, round_number=rnd, model_state=state, metrics={},
                       timestamp=get_current_time())


def good_state(scale=1.0):
    return {"w": torch.full((3, 3), 0.1 * scale), "b": torch.full((3,), 0.1 * scale)}


SHAPES = {"w": torch.Size([3, 3]), "b": torch.Size([3])}


def test_validation_config_defaults():
    cfg = ValidationConfig()
    assert cfg.max_norm == 10.0 and cfg.max_update_size == 100 * 1024 * 1024
    assert cfg.min_clients_for_stats == 5 and cfg.z_score_threshold == 2.0
    assert cfg.signature_required is True


def test_shape_and_range_validation():
    v = DefaultModelValidator(ValidationConfig())
    assert v.validate_shape(upd(good_state()), SHAPES) is ValidationResult.VALID
    assert v.validate_shape(upd({"w": torch.zeros(2, 2), "b": torch.zeros(3)}), SHAPES) \
        is ValidationResult.INVALID_SHAPE
    assert v.validate_shape(upd({"w": torch.zeros(3, 3)}), SHAPES) is ValidationResult.INVALID_SHAPE
    cfg = ValidationConfig()
    assert v.validate_range(upd(good_state()), cfg) is ValidationResult.VALID
    nan = good_state()
    nan["w"][0, 0] = float("nan")
    assert v.validate_range(upd(nan), cfg) is ValidationResult.INVALID_RANGE
    inf = good_state()
    inf["b"][1] = float("inf")
    assert v.validate_range(upd(inf), cfg) is ValidationResult.INVALID_RANGE
    assert v.validate_range(upd(good_state(scale=1000.0)), cfg) is ValidationResult.INVALID_RANGE


def test_statistical_screening():
    v = DefaultModelValidator(ValidationConfig())
    torch.manual_seed(0)
    refs = [upd({k: t + 0.01 * torch.randn_like(t) for k, t in good_state().items()}, cid=str(i))
            for i in range(6)]
    assert v.validate_statistics(upd(good_state()), refs) is ValidationResult.VALID
    assert v.validate_statistics(upd(good_state(scale=50.0)), refs) is ValidationResult.ANOMALOUS
    # too few reference updates: nothing to compare with
    assert v.validate_statistics(upd(good_state(scale=50.0)), refs[:3]) is ValidationResult.VALID
    assert validation_mask([ValidationResult.VALID, ValidationResult.ANOMALOUS]) == [1.0, 0.0]


def test_signatures():
    a, b = SecurityManager(), SecurityManager()
    u = upd(good_state())
    sig = a.sign_update(u)
    assert a.get_public_key().startswith(b"-----BEGIN PUBLIC KEY-----")
    assert a.verify_signature(u, sig, a.get_public_key())
    tampered = upd(good_state(scale=1.01))
    assert not a.verify_signature(tampered, sig, a.get_public_key())
    assert not a.verify_signature(u, sig, b.get_public_key())
    assert not a.verify_signature(u, b"\x00" * len(sig), a.get_public_key())


# -- checkpoints -------------------------------------------------------------
def meta(round_id, state=RoundState.COMPLETED, with_update=False):
    updates = {}
    if with_update:
        updates["client_1"] = upd({"w": torch.ones(2, 2)}, cid="client_1", rnd=round_id)
    return CheckpointMetadata(
        round_id=round_id, timestamp=get_current_time(), num_clients=len(updates),
        client_updates=updates, global_model_version=f"model_v_{round_id}", state=state,
    )


def test_file_state_store_layout_and_round_trip(tmp_path):
    store = FileStateStore(tmp_path)
    store.save_checkpoint(meta(3, with_update=True), {"model_state": {"w": torch.ones(2)}, "n": 3})
    rdir = tmp_path / "checkpoints" / "round_3"
    assert (rdir / "metadata.json").is_file() and (rdir / "state.pt").is_file()
    raw = json.loads((rdir / "metadata.json").read_text())
    assert raw["round_id"] == 3 and raw["state"] == "COMPLETED"
    assert raw["client_updates"]["client_1"]["model_state"]["w"] == [[1.0, 1.0], [1.0, 1.0]]
    loaded = store.load_checkpoint(3)
    assert loaded is not None
    m, state = loaded
    assert m.round_id == 3 and m.state is RoundState.COMPLETED
    assert torch.equal(m.client_updates["client_1"]["model_state"]["w"], torch.ones(2, 2))
    assert torch.equal(state["model_state"]["w"], torch.ones(2))
    assert store.load_checkpoint(99) is None
    store.save_checkpoint(meta(10), {})
    store.save_checkpoint(meta(1), {})
    assert [m.round_id for m in store.list_checkpoints()] == [1, 3, 10]


def test_recovery_strategy():
    s = SimpleRecoveryStrategy()
    for exc in (TimeoutError(), ConnectionError(), RuntimeError("hip")):
        assert s.should_recover(exc)
    assert not s.should_recover(ValueError()) and not s.should_recover(KeyError())
    cps = [meta(1), meta(2, RoundState.FAILED), meta(3, RoundState.IN_PROGRESS)]
    assert s.get_recovery_point(cps).round_id == 1
    assert s.get_recovery_point([meta(5, RoundState.FAILED)]) is None
    assert s.get_recovery_point([]) is None


def test_fault_tolerant_coordinator_rolls_back_to_last_completed(tmp_path):
    ft = FaultTolerantCoordinator(tmp_path)
    ft.checkpoint_round(1, {}, "v1", {"model_state": {"w": torch.full((2,), 1.0)}}, RoundState.COMPLETED)
    ft.checkpoint_round(2, {}, "v2", {"model_state": {"w": torch.full((2,), 2.0)}}, RoundState.FAILED)
    m, state = ft.restore_round(1)
    assert m.global_model_version == "v1"
    recovered = ft.handle_failure(RuntimeError("device lost"), current_round=2)
    assert recovered is not None and recovered[0].round_id == 1
    assert torch.equal(recovered[1]["model_state"]["w"], torch.full((2,), 1.0))
    assert ft.handle_failure(ValueError("bad input"), current_round=2) is None


# -- model versions -----------------------------------------------------------
def test_model_manager_versions_on_disk(tmp_path):
    model = nn.Linear(3, 2)
    mm = ModelManager(model)
    with pytest.raises(ModelManagerError):
        mm.save_model({"x": 1})
    models, configs = tmp_path / "m", tmp_path / "c"
    models.mkdir(); configs.mkdir()
    mm.set_dirs(models, configs)
    boot = mm.list_versions()
    assert len(boot) == 1 and boot[0].config == {"name": "default", "version": "1.0"}
    with torch.no_grad():
        model.weight.fill_(0.25)
    v = mm.save_model({"round_id": 0, "when": get_current_time()}, metrics={"loss": 0.1})
    assert v.version_id.startswith("model_v_") and v.path == models / f"{v.version_id}.pt"
    raw = json.loads((configs / f"{v.version_id}.json").read_text())
    assert raw["version_id"] == v.version_id and raw["config"]["round_id"] == 0
    assert isinstance(raw["config"]["when"], str) and "timestamp" in raw
    state = torch.load(v.path, weights_only=True)
    assert set(state) == {"weight", "bias"} and state["weight"].dtype == torch.float32
    with torch.no_grad():
        model.weight.fill_(9.0)
    latest = mm.load_model()
    assert latest.version_id == v.version_id and mm.current_version.version_id == v.version_id
    assert float(model.weight[0, 0]) == 0.25
    first = mm.load_model(boot[0].version_id)
    assert first.version_id == boot[0].version_id
    with pytest.raises(ModelManagerError):
        mm.load_model("model_v_19990101_000000_001")
    assert [x.version_id for x in mm.list_versions()] == [boot[0].version_id, v.version_id]
    # a second manager over the same directories does not write a bootstrap version
    mm2 = ModelManager(nn.Linear(3, 2))
    mm2.set_dirs(models, configs)
    assert len(mm2.list_versions()) == 2
