"""Every hand-written MNIST-CNN kernel against a plain PyTorch fp32 reference of
the same op.  Each stage is fed the previous stage's *device* output, so a
comparison only sees that stage's own rounding (bf16 storage of activations
and weights, fp32 accumulation)."""

import pytest
import torch
import torch.nn.functional as F

from nanofed_amd.engine import EngineConfig, FederatedEngine
from nanofed_amd.parallel import Communicator

pytestmark = pytest.mark.gpu

G, B = 2, 64


def bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.fixture(scope="module")
def env():
    torch.cuda.set_device(0)
    comm = Communicator(device=torch.device("cuda", 0))
    cfg = EngineConfig(model="mnist_cnn", num_clients=G, samples_per_client=3 * B,
                       batch_size=B, local_epochs=1, eval_samples=200, backend="native",
                       graph_capture=False, learning_rate=0.05)
    eng = FederatedEngine(cfg, comm=comm)
    tr = eng.trainer
    buf = tr.buf
    # distinct, well-scaled random weights per client
    gen = torch.Generator().manual_seed(7)
    sds = []
    for k in range(G):
        sd = {
            "conv1.weight": torch.randn(32, 1, 3, 3, generator=gen) * 0.3,
            "conv1.bias": torch.randn(32, generator=gen) * 0.1,
            "conv2.weight": torch.randn(64, 32, 3, 3, generator=gen) * 0.06,
            "conv2.bias": torch.randn(64, generator=gen) * 0.1,
            "fc1.weight": torch.randn(128, 9216, generator=gen) * 0.01,
            "fc1.bias": torch.randn(128, generator=gen) * 0.1,
            "fc2.weight": torch.randn(10, 128, generator=gen) * 0.1,
            "fc2.bias": torch.randn(10, generator=gen) * 0.1,
        }
        eng.pool[k].load_state_dict(sd)
        sds.append({n: v.cuda() for n, v in sd.items()})
    buf.ctx.p_drop1 = 0.0
    buf.ctx.p_drop2 = 0.0
    buf.ctx.fuse_sgd = 0  # keep fc1.weight's gradient in the arena for the comparisons
    tr._fill_indices(0, 0)
    buf.nvalid[1, 0] = 50  # client 1: partial batch in step 0
    buf.pack()
    torch.cuda.synchronize()
    idx = buf.idx[:, :B].long()
    x = ((buf.images[idx].float() / 255.0 - buf.ctx.mean) * buf.ctx.inv_std).view(G, B, 1, 28, 28)
    y = buf.labels[idx].long()
    return dict(eng=eng, tr=tr, buf=buf, sds=sds, x=x, y=y)


def run(buf, *names):
    for n in names:
        buf.run_stage(n, 0, 0)
    torch.cuda.synchronize()


def test_pack_shadow_and_transposed_conv2(env):
    buf, eng = env["buf"], env["eng"]
    assert torch.equal(buf.shadow, eng.pool.data.to(torch.bfloat16))
    off = eng.layout["conv2.weight"].offset
    for k in range(G):
        w = buf.shadow[k, off : off + 18432].view(64, 9, 32)  # [oc][tap][c]
        assert torch.equal(buf.w2t[k].view(9, 32, 64), w.permute(1, 2, 0))
        assert torch.equal(
            w.view(64, 3, 3, 32).permute(0, 3, 1, 2).float(), bf(env["sds"][k]["conv2.weight"])
        )


def test_conv1_fwd(env):
    buf = env["buf"]
    run(buf, "conv1_fwd")
    for k in range(G):
        ref = F.relu(F.conv2d(env["x"][k], env["sds"][k]["conv1.weight"], env["sds"][k]["conv1.bias"]))
        got = buf.a1[k].float().view(B, 26, 26, 32).permute(0, 3, 1, 2)
        torch.testing.assert_close(got, ref, rtol=1e-2, atol=1e-2)


def test_conv2_fwd_relu_pool(env):
    buf = env["buf"]
    run(buf, "conv1_fwd", "conv2_fwd")
    for k in range(G):
        a1 = buf.a1[k].float().view(B, 26, 26, 32).permute(0, 3, 1, 2)
        pre = F.conv2d(a1, bf(env["sds"][k]["conv2.weight"]), env["sds"][k]["conv2.bias"])
        pooled, arg = F.max_pool2d(F.relu(pre), 2, return_indices=True)
        got = buf.p2[k].float().view(B, 12, 12, 64).permute(0, 3, 1, 2)
        torch.testing.assert_close(got, pooled, rtol=1e-2, atol=2e-2)
        mask = buf.pmask[k].view(B, 12, 12, 64).permute(0, 3, 1, 2).long()
        alive = (mask & 4) > 0
        # alive <=> pooled value positive (allow disagreement only at ~0)
        disagree = alive != (pooled > 0)
        assert float(pooled[disagree].abs().max() if disagree.any() else 0) < 1e-3
        # arg-max position inside the window, where the maximum is clear-cut
        oy, ox = arg // 24, arg % 24
        q_ref = (oy % 2) * 2 + (ox % 2)
        win = F.unfold(F.relu(pre), 2, stride=2).view(B, 64, 4, 144)
        top2 = win.topk(2, dim=2).values
        clear = ((top2[:, :, 0] - top2[:, :, 1]) > 1e-2).view(B, 64, 12, 12) & (pooled > 1e-2)
        assert clear.float().mean() > 0.3
        assert torch.equal((mask & 3)[clear], q_ref[clear])


@pytest.mark.parametrize("split", [1, 3])
def test_fused_conv1_conv2_forward_equals_the_two_kernels(env, split):
    """conv12_fwd (conv1 into the LDS image that conv2 reads, one launch) against
    conv1_fwd followed by conv2_fwd: the same bits in every output, for whole-image and
    for per-band workgroups."""
    buf = env["buf"]
    saved = buf.ctx.split
    buf.ctx.split = split
    try:
        run(buf, "conv1_fwd", "conv2_fwd")
        want = [t.clone() for t in (buf.a1, buf.a1mask, buf.p2, buf.pmask)]
        for t in (buf.a1, buf.a1mask, buf.p2, buf.pmask):
            t.fill_(7)
        run(buf, "conv12_fwd")
        for got, ref, name in zip((buf.a1, buf.a1mask, buf.p2, buf.pmask), want,
                                  ("a1", "a1mask", "p2", "pmask")):
            assert torch.equal(got, ref), name
    finally:
        buf.ctx.split = saved


def _da2_from_pooled(buf, k):
    """d(conv2 out) [B,24,24,64] as the conv2 backward kernels expand it from
    the pooled gradient and the arg-max code."""
    dp2 = buf.dp2[k].float().view(B, 12, 12, 64)
    am = buf.pmask[k].view(B, 12, 12, 64).long() & 3
    out = torch.zeros(B, 24, 24, 64, device="cuda")
    for q in range(4):
        out[:, (q >> 1)::2, (q & 1)::2, :] = (am == q).float() * dp2
    return out


def _native_w1(env, k):
    eng = env["eng"]
    seg = eng.layout["fc1.weight"]
    return env["buf"].shadow[k, seg.offset : seg.offset + seg.numel].float().view(128, 9216)


def test_fc1_fwd_and_head(env):
    buf, eng = env["buf"], env["eng"]
    run(buf, "conv1_fwd", "conv2_fwd", "fc1_fwd", "head", "finalize")
    for k in range(G):
        sd = env["sds"][k]
        p2 = buf.p2[k].float().view(B, 9216)
        h1pre = p2 @ _native_w1(env, k).t() + sd["fc1.bias"]
        got_pre = buf.fc1_slabs[k].sum(0) + sd["fc1.bias"]
        torch.testing.assert_close(got_pre, h1pre, rtol=1e-4, atol=1e-3)
        h1 = F.relu(h1pre).detach().requires_grad_(True)
        w2 = sd["fc2.weight"].clone().requires_grad_(True)
        b2 = sd["fc2.bias"].clone().requires_grad_(True)
        logits = h1 @ w2.t() + b2
        nv = int(buf.nvalid[k, 0])
        loss = F.cross_entropy(logits[:nv], env["y"][k][:nv])
        loss.backward()
        dh1 = h1.grad * (h1pre > 0)
        torch.testing.assert_close(buf.dh1[k].float(), dh1, rtol=2e-2, atol=1e-4)
        assert float(buf.dh1[k, nv:].abs().max() if nv < B else 0) == 0.0
        gv = eng.pool[k].grad_views()
        torch.testing.assert_close(gv["fc2.weight"], w2.grad, rtol=1e-3, atol=1e-5)
        torch.testing.assert_close(gv["fc2.bias"], b2.grad, rtol=1e-3, atol=1e-5)
        # fc1.bias gradient is reduced from the bf16-rounded dh1 path in fp32
        torch.testing.assert_close(gv["fc1.bias"], dh1.sum(0), rtol=1e-3, atol=1e-5)


def test_metrics_accumulate_loss_and_accuracy(env):
    buf = env["buf"]
    buf.metrics.zero_()
    run(buf, "conv1_fwd", "conv2_fwd", "fc1_fwd", "head", "finalize")
    for k in range(G):
        sd = env["sds"][k]
        p2 = buf.p2[k].float().view(B, 9216)
        logits = F.relu(p2 @ _native_w1(env, k).t() + sd["fc1.bias"]) @ sd["fc2.weight"].t() + sd["fc2.bias"]
        nv = int(buf.nvalid[k, 0])
        loss = F.cross_entropy(logits[:nv], env["y"][k][:nv])
        acc = (logits[:nv].argmax(1) == env["y"][k][:nv]).float().mean()
        assert float(buf.metrics[k, 0]) == pytest.approx(float(loss), rel=1e-3)
        assert float(buf.metrics[k, 1]) == pytest.approx(float(acc), abs=1e-6)
        assert float(buf.metrics[k, 2]) == 1.0


def test_fc1_wgrad(env):
    buf, eng = env["buf"], env["eng"]
    run(buf, "conv1_fwd", "conv2_fwd", "fc1_fwd", "head", "fc1_wgrad")
    seg = eng.layout["fc1.weight"]
    for k in range(G):
        ref = buf.dh1[k].float().t() @ buf.p2[k].float().view(B, 9216)
        got = eng.pool.grad[k, seg.offset : seg.offset + seg.numel].view(128, 9216)
        torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-6)


def test_fc1_dgrad_pool_backward(env):
    buf, eng = env["buf"], env["eng"]
    run(buf, "conv1_fwd", "conv2_fwd", "fc1_fwd", "head", "fc1_dgrad", "finalize")
    for k in range(G):
        dp2 = (buf.dh1[k].float() @ _native_w1(env, k)).view(B, 12, 12, 64)
        alive = (buf.pmask[k].view(B, 12, 12, 64).long() & 4) > 0
        want = alive.float() * dp2
        got = buf.dp2[k].float().view(B, 12, 12, 64)
        torch.testing.assert_close(got, want, rtol=1e-2, atol=1e-6)
        # conv2.bias: column sums of the stored values (MFMA against ones inside fc1_dgrad)
        gv = eng.pool[k].grad_views()
        ref_b = got.sum((0, 1, 2))
        torch.testing.assert_close(gv["conv2.bias"], ref_b, rtol=1e-3,
                                   atol=1e-4 * float(ref_b.abs().max()))


def test_conv2_dgrad_with_conv1_wgrad(env):
    buf, eng = env["buf"], env["eng"]
    run(buf, "conv1_fwd", "conv2_fwd", "fc1_fwd", "head", "fc1_dgrad", "conv2_dgrad",
        "conv2_wgrad", "finalize")
    for k in range(G):
        da2 = _da2_from_pooled(buf, k).permute(0, 3, 1, 2)
        a1 = buf.a1[k].float().view(B, 26, 26, 32).permute(0, 3, 1, 2)
        # ReLU mask words: the bit of channel c is (c & 3) + 4 (c >> 3) + 16 ((c >> 2) & 1),
        # the order in which conv1's accumulator registers hold the channels
        ch = torch.arange(32, device="cuda")
        bit = (ch & 3) + 4 * (ch >> 3) + 16 * ((ch >> 2) & 1)
        assert torch.equal(
            (buf.a1mask[k].view(B, 676, 1) >> bit) & 1,
            (buf.a1[k].view(B, 676, 32) > 0).int(),
        )
        da1 = F.conv_transpose2d(da2, bf(env["sds"][k]["conv2.weight"])) * (a1 > 0)
        ref_w = torch.nn.grad.conv2d_weight(env["x"][k], (32, 1, 3, 3), da1)
        gv = eng.pool[k].grad_views()
        scale = float(ref_w.abs().max())
        torch.testing.assert_close(gv["conv1.weight"], ref_w, rtol=1e-2, atol=1e-2 * scale)
        torch.testing.assert_close(gv["conv1.bias"], da1.sum((0, 2, 3)), rtol=1e-2,
                                   atol=1e-2 * float(da1.sum((0, 2, 3)).abs().max()))


def test_conv2_wgrad(env):
    buf, eng = env["buf"], env["eng"]
    run(buf, "conv1_fwd", "conv2_fwd", "fc1_fwd", "head", "fc1_dgrad", "conv2_dgrad",
        "conv2_wgrad", "finalize")
    seg = eng.layout["conv2.weight"]
    for k in range(G):
        da2 = _da2_from_pooled(buf, k).permute(0, 3, 1, 2).contiguous()
        a1 = buf.a1[k].float().view(B, 26, 26, 32).permute(0, 3, 1, 2).contiguous()
        ref = torch.nn.grad.conv2d_weight(a1, (64, 32, 3, 3), da2)
        got = eng.pool.grad[k, seg.offset : seg.offset + seg.numel].view(64, 3, 3, 32).permute(0, 3, 1, 2)
        scale = float(ref.abs().max())
        torch.testing.assert_close(got, ref, rtol=1e-3, atol=1e-4 * scale)


def test_full_step_matches_autograd_of_the_reference_model(env):
    """One whole native step (no dropout) vs autograd on the fp32 nn.Module."""
    from nanofed_amd.models import MNISTModel

    buf, eng = env["buf"], env["eng"]
    before = [eng.pool[k].state_dict("cuda") for k in range(G)]
    buf.train_steps(0, 1)
    torch.cuda.synchronize()
    unfused = eng.pool.data.clone()
    for k in range(G):
        eng.pool[k].load_state_dict(before[k])
    buf.pack()
    buf.ctx.fuse_sgd = 1
    buf.train_steps(0, 1)
    torch.cuda.synchronize()
    buf.ctx.fuse_sgd = 0
    # the fused fc1.weight update is the same arithmetic in a different kernel
    torch.testing.assert_close(eng.pool.data, unfused, rtol=1e-6, atol=1e-7)
    assert torch.equal(buf.shadow, eng.pool.data.to(torch.bfloat16))
    lr = eng.cfg.learning_rate
    for k in range(G):
        model = MNISTModel().cuda()
        model.load_state_dict(before[k])
        model.eval()  # dropout off, same as p_drop = 0
        nv = int(buf.nvalid[k, 0])
        loss = F.nll_loss(model(env["x"][k][:nv]), env["y"][k][:nv])
        loss.backward()
        after = eng.pool[k].state_dict("cuda")
        for name, p in model.named_parameters():
            step = (before[k][name] - after[name]) / lr  # the native gradient
            cos = F.cosine_similarity(step.flatten(), p.grad.flatten(), dim=0)
            assert float(cos) > 0.995, (name, float(cos))
            assert float(step.norm()) == pytest.approx(float(p.grad.norm()), rel=5e-2)
    # restore for the following tests
    for k in range(G):
        eng.pool[k].load_state_dict(before[k])
    buf.pack()


def test_dropout_rates_and_forward_backward_consistency(env):
    buf = env["buf"]
    buf.ctx.p_drop1, buf.ctx.p_drop2 = 0.25, 0.5
    try:
        run(buf, "conv1_fwd", "conv2_fwd", "fc1_fwd", "head", "fc1_dgrad")
        mask = buf.pmask.long()
        positive = buf.p2.float() > 0
        alive = (mask & 4) > 0
        assert torch.equal(alive, positive)
        buf.ctx.p_drop1 = 0.0
        run(buf, "conv2_fwd")
        full = (buf.pmask.long() & 4) > 0
        buf.ctx.p_drop1 = 0.25
        kept = (alive & full).float().sum() / full.float().sum()
        assert float(kept) == pytest.approx(0.75, abs=0.01)
        assert float(buf.dp2.float()[~alive].abs().max()) == 0.0
        # dh1 is zero exactly where fc1's dropout/ReLU killed the unit
        dead = float((buf.dh1.float() == 0).float().mean())
        assert 0.5 < dead < 0.95
    finally:
        buf.ctx.p_drop1, buf.ctx.p_drop2 = 0.0, 0.0


def test_graph_replay_equals_eager_training():
    torch.cuda.set_device(0)
    comm = Communicator(device=torch.device("cuda", 0))
    results = []
    for capture in (False, True):
        cfg = EngineConfig(model="mnist_cnn", num_clients=2, samples_per_client=200,
                           batch_size=B, local_epochs=3, eval_samples=256, backend="native",
                           graph_capture=capture, seed=3)
        eng = FederatedEngine(cfg, comm=comm)
        eng.run(2)
        torch.cuda.synchronize()
        results.append((eng.global_arena.data.clone(), eng.reports[-1].agg_metrics))
    assert torch.equal(results[0][0], results[1][0])
    assert results[0][1] == results[1][1]


def test_native_engine_learns_and_evaluates():
    torch.cuda.set_device(0)
    comm = Communicator(device=torch.device("cuda", 0))
    cfg = EngineConfig(model="mnist_cnn", num_clients=4, samples_per_client=1000,
                       batch_size=B, local_epochs=2, eval_samples=1000, backend="native")
    eng = FederatedEngine(cfg, comm=comm)
    reports = eng.run(3)
    assert all(r.status == "COMPLETED" for r in reports)
    assert reports[-1].agg_metrics["loss"] < reports[0].agg_metrics["loss"]
    ev = eng.evaluate()
    assert ev["accuracy"] > 0.8
    # the native evaluation agrees with the PyTorch model on the exported weights
    from nanofed_amd.models import MNISTModel

    model = MNISTModel().cuda().eval()
    model.load_state_dict(eng.global_state_dict())
    ex = eng.data.eval_inputs.cuda().float().div(255).sub(0.1307).div(0.3081)
    ey = eng.data.eval_labels.cuda()
    with torch.no_grad():
        acc = (model(ex).argmax(1) == ey).float().mean()
    assert float(acc) == pytest.approx(ev["accuracy"], abs=0.02)


def test_idle_clients_are_skipped_without_changing_the_result():
    """Dirichlet shards: clients run out of batches at different steps.  Skipping an
    idle client's workgroups (ctx.ragged) must give the bits of the run that lets
    them compute results nobody uses."""
    from nanofed_amd.engine import EngineConfig, FederatedEngine
    from nanofed_amd.parallel import Communicator

    outs = []
    for skip in (True, False):
        torch.cuda.set_device(0)
        cfg = EngineConfig(model="mnist_cnn", num_clients=4, samples_per_client=200, batch_size=64,
                           local_epochs=2, eval_samples=130, backend="native", seed=11,
                           partition="dirichlet", dirichlet_alpha=0.3)
        eng = FederatedEngine(cfg, comm=Communicator(device=torch.device("cuda", 0)))
        steps = {eng.shards.steps_per_epoch(k) for k in range(4)}
        assert len(steps) > 1, "the shards must be unequal for this test"
        assert eng.trainer.buf.ctx.ragged == 1
        if not skip:
            eng.trainer.buf.ctx.ragged = 0
        reps = eng.run(2)
        ev = eng.evaluate()
        outs.append((eng.global_arena.data.clone(), [r.agg_metrics["loss"] for r in reps], ev))
    assert torch.equal(outs[0][0], outs[1][0])
    assert outs[0][1] == outs[1][1]
    # the evaluation adds per-workgroup sums in a fixed order: identical, not close
    assert outs[0][2] == outs[1][2]


def test_client_groups_train_the_same_model(monkeypatch):
    """Four clients as four concurrent chains against one chain of four.  With
    the kernel geometry pinned every client's partial sums are formed the same way,
    so the models must agree bit for bit."""
    from nanofed_amd.engine import EngineConfig, FederatedEngine
    from nanofed_amd.parallel import Communicator

    monkeypatch.setenv("NANOFED_MNIST_GEOMETRY", "1,32,1")
    outs = []
    for groups in (True, False):
        torch.cuda.set_device(0)
        cfg = EngineConfig(model="mnist_cnn", num_clients=4, samples_per_client=256, batch_size=64,
                           local_epochs=3, eval_samples=64, backend="native", seed=4,
                           client_streams=groups)
        eng = FederatedEngine(cfg, comm=Communicator(device=torch.device("cuda", 0)))
        assert len(eng.trainer._groups) == (4 if groups else 1)
        reps = eng.run(2)   # six epochs: the stream trials and the chosen set
        if groups:
            tr = eng.trainer
            report = tr.stream_report
            assert tr._group_streams and report["streams"] == 4
            if not report["reused"]:  # (an earlier test of this process may have settled it)
                assert len(report["trial_epoch_ms"]) in (tr.STREAM_TRIALS, tr.STREAM_TRIALS_MAX)
            # a second trainer on this device starts on the set that won
            again = FederatedEngine(cfg, comm=Communicator(device=torch.device("cuda", 0)))
            again.run(1)
            assert again.trainer.stream_report["reused"]
            assert again.trainer._group_streams is tr._group_streams
            again.close()
        outs.append((eng.global_arena.data.clone(), [r.agg_metrics["loss"] for r in reps]))
    assert torch.equal(outs[0][0], outs[1][0])
    assert outs[0][1] == outs[1][1]


@pytest.mark.parametrize("clients", [1, 2, 8])
def test_three_backward_kernels_in_one_launch_train_the_same_model(monkeypatch, clients):
    """conv2_dgrad, fc1_wgrad and conv2_wgrad only need what fc1_dgrad left behind and share
    a launch of four-wave workgroups (conv2_wgrad in its four-wave form: taps 0-4 and 5-8
    per wave pair instead of a kernel row per pair).  Every accumulator adds the same
    products in the same order: the models agree bit for bit with the sequence of the pair
    launch and the six-wave conv2_wgrad kernel."""
    from nanofed_amd.engine import EngineConfig, FederatedEngine
    from nanofed_amd.parallel import Communicator

    outs = []
    for triple in ("1", "0"):
        monkeypatch.setenv("NANOFED_MNIST_TRIPLE", triple)
        torch.cuda.set_device(0)
        cfg = EngineConfig(model="mnist_cnn", num_clients=clients, samples_per_client=320,
                           batch_size=64, local_epochs=2, eval_samples=64, backend="native", seed=9,
                           client_streams=False)
        eng = FederatedEngine(cfg, comm=Communicator(device=torch.device("cuda", 0)))
        assert eng.trainer.buf.ctx.bwd_triple == int(triple)
        reps = eng.run(2)
        outs.append((eng.global_arena.data.clone(), [r.agg_metrics for r in reps]))
    assert torch.equal(outs[0][0], outs[1][0])
    assert outs[0][1] == outs[1][1]
    # the default: on, for any number of clients on the GPU
    monkeypatch.delenv("NANOFED_MNIST_TRIPLE")
    eng = FederatedEngine(cfg, comm=Communicator(device=torch.device("cuda", 0)))
    assert eng.trainer.buf.ctx.bwd_triple == 1


def test_make_streams_returns_usable_streams():
    from nanofed_amd.utils.streams import make_streams

    streams = make_streams(3, torch.device("cuda", 0))
    assert len(streams) == 3 and len({s.cuda_stream for s in streams}) == 3
    x = torch.ones(1024, device="cuda")
    outs = []
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            outs.append(x * 2 + 1)
    for s in streams:
        torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    assert all(bool((o == 3).all()) for o in outs)
