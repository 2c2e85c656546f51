"""The Philox noise generators (``privacy/noise/device.py``) on the host evaluation of the
stream: moments, reproducibility, independence of consecutive calls, input validation."""

import math

import pytest
import torch

from nanofed_amd.privacy.noise import DeviceGaussianNoiseGenerator, DeviceLaplacianNoiseGenerator


def test_gaussian_moments_and_shape():
    gen = DeviceGaussianNoiseGenerator(seed=3, device="cpu")
    x = gen.generate((400, 500), scale=2.0)
    assert x.shape == (400, 500) and x.dtype == torch.float32 and x.device.type == "cpu"
    assert abs(float(x.mean())) < 0.02
    assert float(x.std()) == pytest.approx(2.0, rel=0.01)
    # tails as a normal has them: about 4.55 % beyond two sigmas
    assert float((x.abs() > 4.0).float().mean()) == pytest.approx(0.0455, abs=0.003)


def test_laplace_moments():
    gen = DeviceLaplacianNoiseGenerator(seed=5, device="cpu")
    x = gen.generate((200000,), scale=1.5)
    assert abs(float(x.mean())) < 0.02
    assert float(x.abs().mean()) == pytest.approx(1.5, rel=0.02)          # E|x| = b
    assert float(x.std()) == pytest.approx(1.5 * math.sqrt(2.0), rel=0.02)  # sd = b * sqrt(2)


def test_same_seed_same_stream_and_calls_do_not_repeat():
    a = DeviceGaussianNoiseGenerator(seed=11, device="cpu")
    b = DeviceGaussianNoiseGenerator(seed=11, device="cpu")
    a1, a2 = a.generate((1000,), 1.0), a.generate((1000,), 1.0)
    b1, b2 = b.generate((1000,), 1.0), b.generate((1000,), 1.0)
    assert torch.equal(a1, b1) and torch.equal(a2, b2)
    assert not torch.equal(a1, a2)
    assert abs(float(torch.corrcoef(torch.stack([a1, a2]))[0, 1])) < 0.1
    # another seed, another stream; set_seed restarts the stream
    c = DeviceGaussianNoiseGenerator(seed=12, device="cpu")
    assert not torch.equal(c.generate((1000,), 1.0), a1)
    a.set_seed(11)
    assert torch.equal(a.generate((1000,), 1.0), a1)
    # the scale multiplies the same samples
    b.set_seed(11)
    torch.testing.assert_close(b.generate((1000,), 3.0), 3.0 * a1)


def test_unseeded_generators_differ_and_bad_input_is_refused():
    x = DeviceGaussianNoiseGenerator(device="cpu").generate((64,), 1.0)
    y = DeviceGaussianNoiseGenerator(device="cpu").generate((64,), 1.0)
    assert not torch.equal(x, y)
    gen = DeviceGaussianNoiseGenerator(seed=1, device="cpu")
    with pytest.raises(Exception):
        gen.generate((8,), scale=-1.0)
    with pytest.raises(Exception):
        gen.generate((0, -3), scale=1.0)
