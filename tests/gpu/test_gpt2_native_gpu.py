"""GPT-2 native trainer on the GPU: gradients against autograd (fp32 model with
the same bf16-rounded weights) and a federated run that learns."""

import pytest
import torch

from nanofed_amd.engine import EngineConfig, FederatedEngine
from nanofed_amd.engine.gpt2_native import GPT2NativeTrainer
from nanofed_amd.models import GPT2Config, GPT2LM
from nanofed_amd.parallel import Communicator

pytestmark = pytest.mark.gpu
KW = dict(vocab_size=1013, n_positions=128, n_embd=192, n_layer=3, n_head=3)


def _engine(**over):
    cfg = dict(model="gpt2_125m", dataset="synthetic_tokens", num_clients=2, samples_per_client=8,
               batch_size=4, local_epochs=1, seq_len=128, learning_rate=0.05, eval_samples=4,
               model_kwargs=KW)
    cfg.update(over)
    torch.cuda.set_device(0)
    return FederatedEngine(EngineConfig(**cfg), comm=Communicator(device=torch.device("cuda", 0)))


def test_backend_is_native_and_gradients_match_autograd():
    eng = _engine()
    tr = eng.trainer
    assert eng.backend == "native" and isinstance(tr, GPT2NativeTrainer)
    eng.pool.broadcast_from(eng.global_arena.data)
    toks = eng.shards.inputs[:4]
    loss, grads = tr.loss_and_grads(0, toks)
    model = GPT2LM(GPT2Config(**KW)).cuda()
    state = {k: v.to(torch.bfloat16).float() for k, v in eng.global_state_dict().items()}
    model.load_state_dict(state)
    ref = model.loss(toks)
    ref.backward()
    assert abs(loss - ref.item()) < 1e-2 * abs(ref.item())
    named = dict(model.named_parameters())
    for name, g in grads.items():
        want = named[name].grad
        err = float((g.float() - want).norm() / (want.norm() + 1e-8))
        assert err < 6e-2, (name, err)


def test_federated_rounds_learn():
    eng = _engine(learning_rate=0.3, samples_per_client=16)
    first = eng.run_round()
    for _ in range(4):
        last = eng.run_round()
    assert last.status == "COMPLETED"
    assert last.agg_metrics["loss"] < first.agg_metrics["loss"] - 0.1
    ev = eng.evaluate()
    assert ev["loss"] > 0 and ev["samples"] == 4.0
