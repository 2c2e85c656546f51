"""Token streams from files (nanofed_amd/data/tokens.py): windows of a memory-mapped flat
array of ids, the checks that refuse a wrong element type or vocabulary, and the path from
``EngineConfig(dataset="tokens")`` to the clients' shards."""

import numpy as np
import pytest
import torch

from nanofed_amd.data import (
    find_token_file,
    load_token_sequences,
    open_token_stream,
    token_windows,
    write_token_file,
)


def test_windows_are_slices_of_the_stream_and_reproducible(tmp_path):
    stream = np.arange(1000) % 500
    write_token_file(tmp_path / "train.bin", stream)
    assert find_token_file(tmp_path, "train") == tmp_path / "train.bin"
    mapped = open_token_stream(tmp_path / "train.bin")
    assert isinstance(mapped, np.memmap) and mapped.dtype == np.dtype("<u2") and len(mapped) == 1000
    a = load_token_sequences(tmp_path, "train", 16, 20, vocab_size=500, seed=3)
    b = load_token_sequences(tmp_path, "train", 16, 20, vocab_size=500, seed=3)
    c = load_token_sequences(tmp_path, "train", 16, 20, vocab_size=500, seed=4)
    assert a.dtype == torch.int64 and tuple(a.shape) == (20, 17)
    assert torch.equal(a, b) and not torch.equal(a, c)
    starts = a[:, 0].tolist()
    assert len(set(starts)) == 20                       # no window twice
    first_ids = {(16 * k) % 500 for k in range(62)}     # windows start every seq_len tokens
    for row in a.tolist():
        assert row[0] in first_ids
        assert all((row[i + 1] - row[i]) % 500 == 1 for i in range(16))
    # the last token of a window is the first of the next one: every token is a target once
    full = token_windows(np.arange(33), 16, 2, vocab_size=64, seed=0)
    assert full.tolist() == [list(range(0, 17)), list(range(16, 33))]


def test_npy_and_wider_element_types(tmp_path):
    ids = np.array([70000, 5, 6, 7, 8, 9, 10, 11, 12], dtype=np.int64)
    write_token_file(tmp_path / "val.npy", ids, dtype="int64")
    got = load_token_sequences(tmp_path, "val", 4, 2, vocab_size=80000)
    assert got.tolist() == [[70000, 5, 6, 7, 8], [8, 9, 10, 11, 12]]
    with pytest.raises(ValueError, match="do not fit"):
        write_token_file(tmp_path / "train.bin", ids)            # 70000 > uint16
    write_token_file(tmp_path / "train.bin", ids, dtype="uint32")
    assert load_token_sequences(tmp_path, "train", 4, 2, 80000, dtype="uint32").tolist() == got.tolist()
    # read with the wrong element type, the low half of 70000 (4464) leaves a small vocabulary
    with pytest.raises(ValueError, match="vocabulary"):
        load_token_sequences(tmp_path, "train", 4, 2, 1000, dtype="uint16")


def test_refusals(tmp_path):
    with pytest.raises(RuntimeError, match="no token file"):
        load_token_sequences(tmp_path, "train", 8, 2)
    write_token_file(tmp_path / "train.bin", np.arange(40))
    with pytest.raises(ValueError, match="windows"):
        load_token_sequences(tmp_path, "train", 8, 9, vocab_size=64)     # 4 windows only
    with pytest.raises(ValueError, match="vocabulary"):
        load_token_sequences(tmp_path, "train", 8, 2, vocab_size=16)
    (tmp_path / "val.bin").write_bytes(b"\x01\x02\x03")
    with pytest.raises(ValueError, match="whole number"):
        load_token_sequences(tmp_path, "val", 1, 1)
    np.save(tmp_path / "test.npy", np.zeros((3, 3)))
    (tmp_path / "val.bin").unlink()
    with pytest.raises(ValueError, match="one-dimensional integer"):
        load_token_sequences(tmp_path, "val", 1, 1)
    with pytest.raises(ValueError, match="split"):
        find_token_file(tmp_path, "dev")


def test_engine_shards_come_from_the_files(tmp_path):
    from nanofed_amd.engine import EngineConfig
    from nanofed_amd.engine.data import build_federated_data

    rng = np.random.default_rng(0)
    write_token_file(tmp_path / "train.bin", rng.integers(0, 300, size=4000))
    write_token_file(tmp_path / "val.bin", rng.integers(0, 300, size=600))
    cfg = EngineConfig(model="gpt2_125m", dataset="tokens", data_dir=str(tmp_path), num_clients=3,
                       samples_per_client=4, batch_size=2, seq_len=32, eval_samples=5,
                       model_kwargs=dict(vocab_size=300, n_layer=1, n_embd=64, n_head=1,
                                         n_positions=32))
    data = build_federated_data(cfg)
    assert data.kind == "tokens" and tuple(data.inputs.shape) == (12, 33)
    assert tuple(data.eval_inputs.shape) == (5, 33)
    assert sorted(int(i) for s in data.shards for i in s) == list(range(12))
    stream = np.fromfile(tmp_path / "train.bin", dtype="<u2").astype(np.int64)
    for row in data.inputs.numpy():
        hits = [s for s in range(0, 4000 - 32, 32) if np.array_equal(stream[s : s + 33], row)]
        assert len(hits) >= 1
