"""Per-parameter cosine of the grouped ResNet trainer's gradients against (a) fp32 PyTorch
and (b) the per-client bf16 autograd path, on the same weights and batch."""
import torch
import torch.nn.functional as F

from nanofed_amd.engine import EngineConfig, FederatedEngine
from nanofed_amd.models import build_model
from nanofed_amd.parallel import Communicator

DEV, BF = "cuda", torch.bfloat16


def cos(a, b):
    a, b = a.float().flatten(), b.float().flatten()
    return float(torch.dot(a, b) / (a.norm() * b.norm()).clamp(min=1e-30))


torch.cuda.set_device(0)
cfg = EngineConfig(model="resnet18", dataset="synthetic_cifar10", num_clients=2, samples_per_client=64,
                   batch_size=32, local_epochs=1, eval_samples=64, learning_rate=0.05, seed=5,
                   backend="native", graph_capture=False)
eng = FederatedEngine(cfg, comm=Communicator(device=torch.device("cuda", 0)))
tr = eng.trainer
tr.pool.broadcast_from(eng.global_arena.data)
rows = torch.cat([torch.arange(32, device=DEV) + int(tr.shards.offsets[k]) for k in range(2)])
loss, gb = tr.loss_and_grads(rows)
k = 0
ref = build_model("resnet18").to(DEV).float().train()
state = {n: t.to(DEV) for n, t in tr.pool[k].state_dict().items()}
for n, t in state.items():
    if t.dim() >= 2:
        state[n] = t.to(BF).float()
ref.load_state_dict(state)
x, y = tr.shards.batch(rows[:32])
l = F.cross_entropy(ref(x.to(BF).float()), y)
params = dict(ref.named_parameters())
g32 = dict(zip(params, torch.autograd.grad(l, list(params.values()))))
module = tr.modules[k].train()
saved = {n: b.clone() for n, b in module.named_buffers()}
with torch.autocast("cuda", dtype=BF):
    out = module(x.contiguous(memory_format=torch.channels_last))
    lb = F.cross_entropy(out.float(), y)
mp = dict(module.named_parameters())
gbf = dict(zip(mp, torch.autograd.grad(lb, list(mp.values()))))
for n, b in module.named_buffers():
    b.copy_(saved[n])
print("loss native %.5f fp32 %.5f bf16-autograd %.5f" % (float(loss[0]), float(l), float(lb)))
for name in params:
    got = tr.layout.view_of(gb[k], tr.layout[name]).float()
    print("%-28s native~fp32 %.4f  native~bf16 %.4f  bf16~fp32 %.4f  |g| %.3e / %.3e" % (
        name, cos(got, g32[name]), cos(got, gbf[name]), cos(gbf[name], g32[name]),
        float(got.norm()), float(g32[name].norm())))
