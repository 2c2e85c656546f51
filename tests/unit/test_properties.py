"""Property-based tests (hypothesis): the JSON tensor codec, the partitioners
and the flat-arena operations hold their contracts for arbitrary inputs."""

import json

import numpy as np
import torch
from hypothesis import given, settings
from hypothesis import strategies as st

from nanofed_amd import ops
from nanofed_amd.communication.codec import (
    decode_state_json,
    dumps_envelope,
    encode_state_json,
    encode_tensor_json,
    loads_envelope,
)
from nanofed_amd.data.partition import dirichlet_partition, iid_partition, sized_partition

shapes = st.lists(st.integers(1, 5), min_size=0, max_size=4)
finite = st.floats(width=32, allow_nan=False, allow_infinity=False)


@settings(max_examples=60, deadline=None)
@given(shape=shapes, seed=st.integers(0, 2**31 - 1), scale=st.sampled_from([1e-30, 1e-3, 1.0, 1e6, 1e30]))
def test_codec_round_trips_every_float32_exactly(shape, seed, scale):
    g = torch.Generator().manual_seed(seed)
    t = torch.randn(tuple(shape), generator=g) * scale
    text = encode_tensor_json(t)
    assert torch.equal(torch.tensor(json.loads(text), dtype=torch.float32).reshape(t.shape), t)
    state = decode_state_json(encode_state_json({"w": t, "b": t.reshape(-1)}))
    assert torch.equal(state["w"].reshape(t.shape), t) and torch.equal(state["b"], t.reshape(-1))


@settings(max_examples=40, deadline=None)
@given(values=st.lists(finite, min_size=1, max_size=40), round_no=st.integers(0, 10**6),
       name=st.text(alphabet=st.characters(blacklist_categories=("Cs",)), max_size=12))
def test_envelope_keeps_scalars_strings_and_tensors(values, round_no, name):
    t = torch.tensor(values, dtype=torch.float32)
    body = dumps_envelope({"client_id": name, "round_number": round_no,
                           "model_state": {"layer.weight": t}, "metrics": {"loss": 0.5}})
    back = loads_envelope(body)
    assert back["client_id"] == name and back["round_number"] == round_no
    assert torch.equal(torch.as_tensor(back["model_state"]["layer.weight"], dtype=torch.float32), t)
    assert back["metrics"] == {"loss": 0.5}


@settings(max_examples=40, deadline=None)
@given(n=st.integers(1, 400), clients=st.integers(1, 12), seed=st.integers(0, 1000))
def test_partitions_cover_every_sample_exactly_once(n, clients, seed):
    parts = iid_partition(n, clients, seed=seed)
    flat = np.concatenate(parts)
    assert len(parts) == clients and sorted(flat.tolist()) == list(range(n))
    assert max(map(len, parts)) - min(map(len, parts)) <= 1
    labels = np.random.default_rng(seed).integers(0, 10, size=n)
    dparts = dirichlet_partition(labels, clients, alpha=0.3, seed=seed, min_size=0)
    dflat = np.concatenate([p for p in dparts if len(p)]) if n else np.zeros(0)
    assert sorted(dflat.tolist()) == list(range(n))


@settings(max_examples=30, deadline=None)
@given(sizes=st.lists(st.integers(0, 50), min_size=1, max_size=8), seed=st.integers(0, 1000))
def test_sized_partition_respects_the_requested_sizes(sizes, seed):
    total = sum(sizes)
    parts = sized_partition(total, sizes, seed=seed)
    assert [len(p) for p in parts] == sizes
    assert sorted(np.concatenate(parts).tolist() if total else []) == list(range(total))


@settings(max_examples=40, deadline=None)
@given(k=st.integers(1, 6), n=st.integers(1, 300), seed=st.integers(0, 10**6))
def test_weighted_sum_and_norms_match_plain_torch(k, n, seed):
    g = torch.Generator().manual_seed(seed)
    rows = [torch.randn(n, generator=g) for _ in range(k)]
    w = torch.rand(k, generator=g)
    w[0] = 0.0  # a masked-out client
    rows[0][0] = float("nan")  # ... whose row may hold anything
    got = ops.weighted_sum(rows, w)
    want = sum(float(w[i]) * rows[i] for i in range(1, k)) if k > 1 else torch.zeros(n)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)
    sq, bad = ops.finite_and_sqnorm(rows[-1] if k > 1 else torch.ones(n))
    ref = rows[-1] if k > 1 else torch.ones(n)
    assert bad == 0 and abs(sq - float((ref.double() ** 2).sum())) <= 1e-4 * max(1.0, sq)


@settings(max_examples=25, deadline=None)
@given(n=st.integers(1, 500), cut=st.integers(0, 500), seed=st.integers(0, 2**40), offset=st.integers(0, 2**40),
       mode=st.sampled_from([ops.MODE_UNIFORM, ops.MODE_NORMAL, ops.MODE_LAPLACE, ops.MODE_SYM_UNIFORM]))
def test_philox_stream_is_a_pure_function_of_seed_offset_and_index(n, cut, seed, offset, mode):
    cut = min(cut, n)
    whole = ops.philox_fill(torch.empty(n), seed=seed, offset=offset, mode=mode)
    parts = torch.empty(n)
    if cut:
        ops.philox_fill(parts[:cut], seed=seed, offset=offset, mode=mode, first=0)
    if cut < n:
        ops.philox_fill(parts[cut:], seed=seed, offset=offset, mode=mode, first=cut)
    assert torch.equal(whole, parts)
    assert torch.isfinite(whole).all()


def test_codec_is_bit_exact_on_every_kind_of_float32_and_parses_like_python():
    """Random bit patterns (subnormals, extremes, negative zero), the non-finite values, and
    arbitrary decimal text: the unchecked writer and the short-cut number parser give what the
    general paths give."""
    import random
    import struct

    from nanofed_amd.communication.codec import _decode_prefix

    rng = random.Random(7)
    values = [0.0, -0.0, 1.0, -1.0, 1e-45, 1.17549435e-38, 3.4028235e38, -3.4028235e38, 0.1,
              1 / 3, 123456.789, 1e10, 1e-10, 16777216.0, float("nan"), float("inf"), -float("inf")]
    values += [struct.unpack("f", struct.pack("I", rng.getrandbits(32)))[0] for _ in range(50000)]
    t = torch.tensor(values, dtype=torch.float32)
    bits = t.view(torch.int32)
    nan = torch.isnan(t)
    text = encode_tensor_json(t)
    ref = torch.tensor([float(x) for x in json.loads(text, parse_constant=float)],
                       dtype=torch.float32)
    assert bool(((ref.view(torch.int32) == bits) | nan).all())
    back, end = _decode_prefix(text, 0)
    assert end == len(text) and bool(((back.view(torch.int32) == bits) | nan).all())
    assert bool((torch.isnan(back) == nan).all())
    # decimal text that no float32 printer wrote: 1-17 digits, exponents far out
    texts = ["%.*g" % (rng.randint(1, 17), rng.uniform(-1, 1) * 10 ** rng.randint(-40, 39))
             for _ in range(20000)] + ["1e-50", "-1e60", "0.000000000000000000001", "12345678901234567890"]
    blob = ("[" + ", ".join(texts) + "]").encode()
    got, _ = _decode_prefix(blob, 0)
    want = torch.tensor([float(x) for x in texts], dtype=torch.float64).to(torch.float32)
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))


def test_codec_writes_the_lists_python_would_for_degenerate_shapes():
    for shape in [(5, 1, 1), (3, 0), (0,), (2, 2, 2, 1), (1, 1, 1, 1, 3), ()]:
        n = int(np.prod(shape)) if shape else 1
        x = torch.arange(n, dtype=torch.float32).reshape(shape)
        assert json.loads(encode_tensor_json(x)) == x.tolist(), shape
    # a 0-d entry (BatchNorm's num_batches_tracked) is a plain number on the wire, as
    # tensor.tolist() makes it, and comes back 0-d
    body = dumps_envelope({"model_state": {"w": torch.ones(2, 2), "n": torch.tensor(7)}})
    assert b'"n": 7.0' in body
    back = loads_envelope(body)
    assert back["model_state"]["n"].shape == () and float(back["model_state"]["n"]) == 7.0
