"""Local trainer of the GPT-2 configuration with an explicit forward/backward.

The model is ``nanofed_amd.models.gpt2.GPT2LM`` (the reference has no sequence
model; its training step is nanofed/trainer/base.py:141-159: zero_grad,
forward, loss, backward, SGD).  Instead of autograd under bf16 autocast this
trainer runs the step by hand on flat buffers:

* fp32 master weights stay in the client's arena row; the forward and backward
  passes read a bf16 copy with the same offsets, refreshed by the SGD kernel, so
  no weight is cast per step;
* gradients are written as bf16 straight into one flat gradient buffer by the
  weight-gradient GEMMs, then consumed by one SGD launch;
* everything between the GEMMs is a hand-written kernel
  (``nanofed_amd.ops.nn_ops`` / ``nanofed_amd.ops.attention``): embedding
  gather and its scatter-free backward, residual-add + LayerNorm, tanh-GELU,
  bias gradients, softmax cross-entropy that overwrites the logits with their
  gradient (the logits are never upcast), causal flash attention;
* GEMMs run on the hand-written MFMA kernels of ``nanofed_amd.ops.gemm_ops``
  (forward, data-gradient and weight-gradient layouts without transposed copies,
  bias and GELU in the forward epilogue); ``NANOFED_NATIVE_GEMM=0`` routes them to
  hipBLASLt through ``torch.mm`` / ``torch.addmm`` instead, for comparison.

Activations are bf16, statistics and losses fp32.

Clients.  With equal shards the clients of a GPU step together (``grouped``): their
batches are stacked, every GEMM is ONE batched launch in which workgroup z reads
client z's weights (``nf_gemm_bf16_batched``: a 4096-row layer GEMM alone makes 1.1
rounds of workgroups and needs split-K for its weight gradient; eight of them fill the
chip evenly - 233 us instead of 361 us per layer and client), attention, the
cross-entropy and SGD are one launch for all clients, LayerNorm and the bias-gradient
column sums take the client from the row index, and only the embedding kernels (one table
per client) are issued per client on row slices.  Otherwise (unequal shards,
``NANOFED_GPT2_GROUP=0``, lanes) the clients are trained one after the other and share
the activation and gradient buffers.
"""

from __future__ import annotations

import os

import torch
import torch.nn.functional as F

from nanofed_amd.core.exceptions import DeviceError
from nanofed_amd.models.gpt2 import GPT2Config
from nanofed_amd.ops import attention as attn_ops
from nanofed_amd.ops import gemm_ops
from nanofed_amd.ops import group_ops as go
from nanofed_amd.ops import nn_ops
from nanofed_amd.ops.group_ops import Ref
from nanofed_amd.runtime.arena import ArenaLayout, ArenaPool
from nanofed_amd.utils.streams import side_stream

from .config import EngineConfig
from .data import DeviceShards, FederatedData
from .trainers import M_ACC, M_LOSS, M_SAMPLES, M_VALID, METRIC_COLUMNS, GroupTrainer

BF16 = torch.bfloat16


# concurrent client steps of the profiled shape (see GPT2NativeTrainer.__init__)
LANES = int(os.environ.get("NANOFED_GPT2_LANES", "1"))
# clients stepped together (0 = one after the other); activations grow with the group
GROUP = int(os.environ.get("NANOFED_GPT2_GROUP", "8"))
# grouped step: every weight-gradient GEMM on a second stream beside the data-gradient GEMM
# that reads the same d(y) (0 = one stream)
WGRAD_STREAM = os.environ.get("NANOFED_GPT2_WGRAD_STREAM", "1") != "0"


def _views(layout: ArenaLayout, flat: torch.Tensor) -> dict[str, torch.Tensor]:
    return {s.name: flat[s.offset : s.offset + s.numel].view(s.shape) for s in layout.segments}


class _Buffers:
    """Activations of one step for ``R = batch * seq`` rows."""

    def __init__(self, m: GPT2Config, batch: int, seq: int, device: torch.device) -> None:
        R, C, L = batch * seq, m.n_embd, m.n_layer
        self.R, self.batch, self.seq = R, batch, seq
        self.ld = (m.vocab_size + 63) // 64 * 64

        def bf(*shape):
            return torch.empty(*shape, dtype=BF16, device=device)

        def f32(*shape):
            return torch.empty(*shape, dtype=torch.float32, device=device)

        self.x = bf(L + 1, R, C)       # input of every block, and of ln_f
        self.x_mid = bf(L, R, C)       # after the attention residual
        self.ln1, self.ln2 = bf(L, R, C), bf(L, R, C)
        self.qkv = bf(L, R, 3 * C)
        self.att = bf(L, R, C)
        self.lse = f32(L, batch, m.n_head, seq)
        self.fc, self.act = bf(L, R, 4 * C), bf(L, R, 4 * C)
        self.stats = f32(2 * L + 1, 2, R)  # mean / rstd of ln_1, ln_2 per layer and ln_f
        self.lnf = bf(R, C)
        self.tmp_c = bf(R, C)
        self.logits = bf(R, self.ld)
        self.loss_rows, self.correct_rows = f32(R), f32(R)
        # backward scratch
        self.dres, self.dln, self.datt = bf(R, C), bf(R, C), bf(R, C)
        self.dact = bf(R, 4 * C)
        self.dqkv = bf(R, 3 * C)
        self.attn_scratch = attn_ops.backward_scratch(batch, m.n_head, seq, device)


class GPT2NativeTrainer(GroupTrainer):
    def __init__(
        self,
        cfg: EngineConfig,
        pool: ArenaPool,
        shards: DeviceShards,
        data: FederatedData,
        model_cfg: GPT2Config,
    ) -> None:
        if cfg.dp is not None:
            raise DeviceError("the GPT-2 trainer has no DP-SGD path")
        if model_cfg.dropout != 0.0:
            raise DeviceError("the GPT-2 trainer does not implement dropout")
        if model_cfg.n_embd % model_cfg.n_head or (model_cfg.n_embd // model_cfg.n_head) != 64:
            raise DeviceError("the attention kernels need a head dimension of 64")
        self.cfg, self.pool, self.shards, self.m = cfg, pool, shards, model_cfg
        self.layout = pool.layout
        self.device = pool.data.device
        G = len(pool)
        n = self.layout.numel
        self.shadow = torch.zeros(G, n, dtype=BF16, device=self.device)
        # Lanes: independent sets of activation / gradient buffers, one HIP stream
        # each; client k trains on lane k % lanes, so steps of different clients
        # overlap on the GPU (many kernels of a 4096-row step leave compute units
        # idle).  With the GEMMs on the vendor library two lanes measured 3.45 -> 4.73
        # rounds/s, but FOUR lanes hung the process twice (no GPU fault) and the
        # cause was never shown - the library's stream-K / split-K GEMM kernels,
        # whose workgroups wait for each other through memory, are the suspects, and
        # they are issued from one host thread that shares one library handle.
        # Every multi-lane configuration runs that same unexplained path, so more
        # than one lane is refused while library GEMMs are in the step; with the
        # hand-written GEMM kernels (no workgroup of which ever waits for another)
        # two lanes are allowed again.
        want = max(LANES, cfg.gpt2_lanes)
        self.native_gemm = gemm_ops.enabled(self.device)
        if want > 2 or (want > 1 and not self.native_gemm):
            raise DeviceError(
                "GPT-2 lanes > 1 are refused: concurrent steps on library GEMM kernels hung "
                "the GPU queue and the cause is open (two lanes need the native GEMM path)"
            )
        lanes = 1
        full_batches = all(sz % cfg.batch_size == 0 for sz in shards.sizes)
        if cfg.client_streams and self.device.type == "cuda" and G > 1 and full_batches:
            lanes = min(G, max(want, 1))
        self.lanes = max(1, lanes)
        self._grads = [torch.zeros(n, dtype=BF16, device=self.device) for _ in range(self.lanes)]
        self._gvs = [_views(self.layout, g) for g in self._grads]
        self._streams = (
            [side_stream(self.device) for _ in range(self.lanes)]
            if self.lanes > 1 else None
        )
        self._eval_shadow = torch.zeros(n, dtype=BF16, device=self.device)
        self._bufs: dict[tuple[int, int, int], _Buffers] = {}
        self._eval_inputs = data.eval_inputs
        # the kernels index the embedding table with the raw ids: check them once
        for toks in (shards.inputs, data.eval_inputs):
            if toks is not None and toks.numel() and (
                int(toks.max()) >= model_cfg.vocab_size or int(toks.min()) < 0
            ):
                raise DeviceError("token id outside the vocabulary")
        self.metrics = torch.zeros(G, METRIC_COLUMNS, dtype=torch.float32, device=self.device)
        self._acc = torch.zeros(G, 3, dtype=torch.float32, device=self.device)
        # grouped steps: equal shards of full batches, one lane, the native GEMM path
        self.group = 0
        if (GROUP > 1 and G > 1 and self.lanes == 1 and self.native_gemm and full_batches
                and self.device.type == "cuda" and len(set(shards.sizes)) == 1):
            self.group = min(GROUP, G)
            self._ggrads = torch.zeros(self.group, n, dtype=BF16, device=self.device)
            self._gslabs = torch.empty(1, dtype=torch.float32, device=self.device)
            self._gslabs_side = torch.empty(1, dtype=torch.float32, device=self.device)
            self._side = (torch.cuda.Stream(self.device)
                          if WGRAD_STREAM and self.device.type == "cuda" else None)
            self.stream_report = {"wgrad_stream": int(self._side is not None)}
        self._host_tokens: torch.Tensor | None = None
        self._step_graphs: dict[tuple[int, int], torch.cuda.CUDAGraph] = {}
        self._step_inputs: dict[tuple[int, int], dict] = {}

    # -- buffers ------------------------------------------------------------
    def _buffers(self, batch: int, seq: int, lane: int = 0) -> _Buffers:
        key = (batch, seq, lane)
        if key not in self._bufs:
            self._bufs[key] = _Buffers(self.m, batch, seq, self.device)
        return self._bufs[key]

    # -- linear layers ------------------------------------------------------
    def _lin_fwd(self, x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None,
                 out: torch.Tensor, gelu_out: torch.Tensor | None = None) -> None:
        """out = x w^T (+ bias); gelu_out = gelu(out) when given."""
        if self.native_gemm:
            gemm_ops.linear_fwd(x, w, bias, out, gelu_out)
            return
        if bias is not None:
            torch.addmm(bias, x, w.t(), out=out)
        else:
            torch.mm(x, w.t(), out=out)
        if gelu_out is not None:
            nn_ops.gelu_fwd(out, gelu_out)

    def _lin_dgrad(self, dy: torch.Tensor, w: torch.Tensor, out: torch.Tensor) -> None:
        if self.native_gemm:
            gemm_ops.linear_dgrad(dy, w, out)
        else:
            torch.mm(dy, w, out=out)

    def _lin_wgrad(self, dy: torch.Tensor, x: torch.Tensor, out: torch.Tensor) -> None:
        if self.native_gemm:
            gemm_ops.linear_wgrad(dy, x, out)
        else:
            torch.mm(dy.t(), x, out=out)

    # -- forward ------------------------------------------------------------
    def _forward(self, w: dict[str, torch.Tensor], tokens: torch.Tensor, b: _Buffers) -> None:
        """Fills the activation buffers and leaves d(logits) in ``b.logits``,
        per-row losses in ``b.loss_rows``."""
        m, L = self.m, self.m.n_layer
        inp = tokens[:, :-1].contiguous().view(-1)
        tgt = tokens[:, 1:].contiguous().view(-1)
        V = m.vocab_size
        nn_ops.embed_fwd(inp, w["wte.weight"], w["wpe.weight"], b.seq, out=b.x[0])
        delta = None
        for l in range(L):
            p = f"h.{l}."
            st1, st2 = b.stats[2 * l], b.stats[2 * l + 1]
            if l == 0:
                nn_ops.ln_fwd(b.x[0], w[p + "ln_1.weight"], w[p + "ln_1.bias"], b.ln1[0],
                              st1[0], st1[1])
            else:
                nn_ops.ln_fwd(b.x_mid[l - 1], w[p + "ln_1.weight"], w[p + "ln_1.bias"],
                              b.ln1[l], st1[0], st1[1], delta=delta, x_out=b.x[l])
            self._lin_fwd(b.ln1[l], w[p + "attn.c_attn.weight"], w[p + "attn.c_attn.bias"], b.qkv[l])
            attn_ops.causal_attention_fwd(b.qkv[l], b.att[l], b.lse[l], b.batch, b.seq, m.n_head)
            self._lin_fwd(b.att[l], w[p + "attn.c_proj.weight"], w[p + "attn.c_proj.bias"], b.tmp_c)
            nn_ops.ln_fwd(b.x[l], w[p + "ln_2.weight"], w[p + "ln_2.bias"], b.ln2[l],
                          st2[0], st2[1], delta=b.tmp_c, x_out=b.x_mid[l])
            self._lin_fwd(b.ln2[l], w[p + "c_fc.weight"], w[p + "c_fc.bias"], b.fc[l],
                          gelu_out=b.act[l])
            self._lin_fwd(b.act[l], w[p + "c_proj.weight"], w[p + "c_proj.bias"], b.tmp_c)
            delta = b.tmp_c
        stf = b.stats[2 * L]
        nn_ops.ln_fwd(b.x_mid[L - 1], w["ln_f.weight"], w["ln_f.bias"], b.lnf, stf[0], stf[1],
                      delta=delta, x_out=b.x[L])
        self._lin_fwd(b.lnf, w["wte.weight"], None, b.logits[:, :V])
        nn_ops.cross_entropy_fwd_bwd(b.logits, tgt, V, 1.0 / b.R, b.loss_rows, b.correct_rows)

    # -- backward -----------------------------------------------------------
    def _backward(self, w: dict[str, torch.Tensor], tokens: torch.Tensor, b: _Buffers,
                  g: dict[str, torch.Tensor]) -> None:
        m, L = self.m, self.m.n_layer
        V = m.vocab_size
        dlog = b.logits[:, :V]
        self._lin_wgrad(dlog, b.lnf, g["wte.weight"])  # LM head shares the embedding matrix
        self._lin_dgrad(dlog, w["wte.weight"], b.dln)
        stf = b.stats[2 * L]
        # every LayerNorm backward also sums its d(x) over the rows: that is the bias
        # gradient of the projection that wrote into the residual stream before it
        nn_ops.ln_bwd(b.dln, b.x[L], w["ln_f.weight"], stf[0], stf[1], b.dres,
                      g["ln_f.weight"], g["ln_f.bias"], dxsum=g[f"h.{L - 1}.c_proj.bias"])
        for l in reversed(range(L)):
            p = f"h.{l}."
            st1, st2 = b.stats[2 * l], b.stats[2 * l + 1]
            # MLP: x_next = x_mid + c_proj(gelu(c_fc(ln_2(x_mid))))
            self._lin_wgrad(b.dres, b.act[l], g[p + "c_proj.weight"])
            self._lin_dgrad(b.dres, w[p + "c_proj.weight"], b.dact)
            nn_ops.gelu_bwd_colsum(b.fc[l], b.dact, b.dact, g[p + "c_fc.bias"])
            self._lin_wgrad(b.dact, b.ln2[l], g[p + "c_fc.weight"])
            self._lin_dgrad(b.dact, w[p + "c_fc.weight"], b.dln)
            nn_ops.ln_bwd(b.dln, b.x_mid[l], w[p + "ln_2.weight"], st2[0], st2[1], b.dres,
                          g[p + "ln_2.weight"], g[p + "ln_2.bias"], dres=b.dres,
                          dxsum=g[p + "attn.c_proj.bias"])
            # attention: x_mid = x + c_proj(attn(c_attn(ln_1(x))))
            self._lin_wgrad(b.dres, b.att[l], g[p + "attn.c_proj.weight"])
            self._lin_dgrad(b.dres, w[p + "attn.c_proj.weight"], b.datt)
            attn_ops.causal_attention_bwd(b.qkv[l], b.att[l], b.lse[l], b.datt, b.dqkv,
                                          b.attn_scratch, b.batch, b.seq, m.n_head)
            nn_ops.colsum(b.dqkv, g[p + "attn.c_attn.bias"])
            self._lin_wgrad(b.dqkv, b.ln1[l], g[p + "attn.c_attn.weight"])
            self._lin_dgrad(b.dqkv, w[p + "attn.c_attn.weight"], b.dln)
            nn_ops.ln_bwd(b.dln, b.x[l], w[p + "ln_1.weight"], st1[0], st1[1], b.dres,
                          g[p + "ln_1.weight"], g[p + "ln_1.bias"], dres=b.dres,
                          dxsum=g[f"h.{l - 1}.c_proj.bias"] if l > 0 else None)
        nn_ops.embed_bwd(tokens[:, :-1].contiguous(), b.dres, g["wte.weight"], g["wpe.weight"],
                         b.batch, b.seq)

    # -- one optimisation step of client k ------------------------------------
    def _step(self, k: int, tokens: torch.Tensor, lane: int = 0) -> None:
        cfg = self.cfg
        b = self._buffers(tokens.shape[0], tokens.shape[1] - 1, lane)
        w = _views(self.layout, self.shadow[k])
        self._forward(w, tokens, b)
        self._backward(w, tokens, b, self._gvs[lane])
        nn_ops.sgd_step_bf16g(self.pool[k].data, self._grads[lane], self.shadow[k],
                              cfg.learning_rate, cfg.weight_decay)
        acc = self._acc[k]
        acc[0] += b.loss_rows.mean()
        acc[1] += b.correct_rows.mean()
        acc[2] += 1.0

    def loss_and_grads(self, k: int, tokens: torch.Tensor) -> tuple[float, dict[str, torch.Tensor]]:
        """Loss and the flat gradient views of one batch, without an update
        (the kernel tests compare these with autograd)."""
        b = self._buffers(tokens.shape[0], tokens.shape[1] - 1)
        self.shadow[k].copy_(self.pool[k].data)
        w = _views(self.layout, self.shadow[k])
        self._forward(w, tokens, b)
        self._backward(w, tokens, b, self._gvs[0])
        return float(b.loss_rows.mean()), self._gvs[0]

    # -- grouped step: clients g0 .. g0 + Gn - 1 together ---------------------------------
    def _off(self, name: str) -> int:
        return self.layout[name].offset

    def _gemm(self, layout: int, a: torch.Tensor, b: "torch.Tensor | Ref", c: "torch.Tensor | Ref",
              M: int, N: int, K: int, lda: int, ldb: int, ldc: int, Gn: int, sa: int, sb: int,
              sc: int, bias: Ref | None = None, c2: torch.Tensor | None = None,
              side: bool = False) -> None:
        # (the side stream has split-K slabs of its own: its GEMM runs beside one of the
        # main stream)
        which = "_gslabs_side" if side else "_gslabs"
        need = go.gemm_slab_floats(M, N, K, layout, Gn)
        if need > getattr(self, which).numel():
            setattr(self, which, torch.empty(need, dtype=torch.float32, device=self.device))
        go.gemm_batched(layout, a, b, c, M, N, K, lda, ldb, ldc, Gn, sa, sb, sc,
                        getattr(self, which), go._stream(a), bias=bias, c2=c2)

    def _glin_fwd(self, x: torch.Tensor, wname: str, bname: str | None, out: torch.Tensor,
                  g0: int, Gn: int, R: int, gelu_out: torch.Tensor | None = None,
                  n_out: int | None = None) -> None:
        """out[g] = x[g] w[g]^T (+ bias[g]) for every client of the group; x ``[Gn * R, K]``,
        out ``[Gn * R, >= N]`` (rows may be longer than N: the padded logits)."""
        N, K = self.layout[wname].shape
        w = Ref(self.shadow[g0:], self._off(wname))
        bias = Ref(self.shadow[g0:], self._off(bname)) if bname is not None else None
        self._gemm(gemm_ops.NT, x, w, out, R, n_out or N, K, x.stride(0), K,
                   out.stride(0), Gn, R * x.stride(0), w.stride, R * out.stride(0), bias, gelu_out)

    def _glin_dgrad(self, dy: torch.Tensor, wname: str, out: torch.Tensor, g0: int, Gn: int,
                    R: int, n_in: int | None = None) -> None:
        """out[g] = dy[g] w[g]; dy ``[Gn * R, >= N]``, out ``[Gn * R, K]``."""
        N, K = self.layout[wname].shape
        w = Ref(self.shadow[g0:], self._off(wname))
        self._gemm(gemm_ops.NN, dy, w, out, R, K, n_in or N, dy.stride(0), K,
                   out.stride(0), Gn, R * dy.stride(0), w.stride, R * out.stride(0))

    def _glin_wgrad(self, dy: torch.Tensor, x: torch.Tensor, wname: str, Gn: int, R: int,
                    n_out: int | None = None, side: bool = False) -> None:
        """grad(w)[g] = dy[g]^T x[g], written into the group's gradient rows."""
        N, K = self.layout[wname].shape
        dw = Ref(self._ggrads, self._off(wname))
        self._gemm(gemm_ops.TN, dy, x, dw, n_out or N, K, R, dy.stride(0),
                   x.stride(0), K, Gn, R * dy.stride(0), R * x.stride(0), dw.stride, side=side)

    def _glin_bwd(self, dy: torch.Tensor, x: torch.Tensor, wname: str, dx: torch.Tensor, g0: int,
                  Gn: int, R: int, n: int | None = None, also=None) -> None:
        """Both gradients of a linear layer (``also``: more work that only reads d(y), e.g.
        the bias gradient; it goes where the weight gradient goes).  They read the same d(y) and write different
        buffers, and neither fills the chip to the last round of workgroups (36-288 tiles of
        256 x 256 per launch on 256 CUs): the weight gradient goes to a second stream of the
        same priority and the main stream joins it behind the data gradient, so that nothing
        else on the main stream sees a difference.  (The c_fc pair of 8 clients: 440 us one
        after the other, 410-425 us side by side, whichever fresh stream carries the second;
        the GPT-2 configuration 5.38 -> 5.47 rounds/s, means of 5 and 7 alternating runs that
        spread over +-3 %.)"""
        side = self._side
        if side is None:
            if also is not None:
                also()
            self._glin_wgrad(dy, x, wname, Gn, R, n_out=n)
            self._glin_dgrad(dy, wname, dx, g0, Gn, R, n_in=n)
            return
        main = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(main)
        side.wait_event(ready)
        with torch.cuda.stream(side):
            if also is not None:
                also()
            self._glin_wgrad(dy, x, wname, Gn, R, n_out=n, side=True)
            done = torch.cuda.Event()
            done.record(side)
        self._glin_dgrad(dy, wname, dx, g0, Gn, R, n_in=n)
        main.wait_event(done)

    def _gstep(self, g0: int, Gn: int, tokens: torch.Tensor,
               order: torch.Tensor | None = None) -> None:
        """One optimisation step of Gn clients; ``tokens`` ``[Gn * batch, seq + 1]``,
        client-major.  ``order`` ``[Gn, 2, batch * seq]``: per client the sorted input tokens
        and their stable sort order (drawn on the host with the batch); without it the
        embedding gradient sorts on the device."""
        cfg, m, L = self.cfg, self.m, self.m.n_layer
        nb = tokens.shape[0] // Gn                       # sequences per client
        b = self._buffers(tokens.shape[0], tokens.shape[1] - 1, lane=100 + Gn)
        R, V = nb * b.seq, m.vocab_size                   # rows per client
        ws = [_views(self.layout, self.shadow[g0 + j]) for j in range(Gn)]
        gs = [_views(self.layout, self._ggrads[j]) for j in range(Gn)]
        rows = [slice(j * R, (j + 1) * R) for j in range(Gn)]
        inp = tokens[:, :-1].contiguous()
        tgt = tokens[:, 1:].contiguous().view(-1)
        flat_inp = inp.view(-1)

        def wref(name):      # bf16 parameter copy of the group's first client
            return Ref(self.shadow[g0:], self._off(name))

        def gref(name):      # gradient slot
            return Ref(self._ggrads, self._off(name))

        def ln(x, name, out, st, delta=None, x_out=None):
            nn_ops.ln_fwd_g(x, wref(name + ".weight"), wref(name + ".bias"), out, st[0], st[1], Gn,
                            delta=delta, x_out=x_out)

        # forward
        for j in range(Gn):
            nn_ops.embed_fwd(flat_inp[rows[j]], ws[j]["wte.weight"], ws[j]["wpe.weight"], b.seq,
                             out=b.x[0][rows[j]])
        delta = None
        for l in range(L):
            p = f"h.{l}."
            st1, st2 = b.stats[2 * l], b.stats[2 * l + 1]
            if l == 0:
                ln(b.x[0], p + "ln_1", b.ln1[0], st1)
            else:
                ln(b.x_mid[l - 1], p + "ln_1", b.ln1[l], st1, delta=delta, x_out=b.x[l])
            self._glin_fwd(b.ln1[l], p + "attn.c_attn.weight", p + "attn.c_attn.bias", b.qkv[l],
                           g0, Gn, R)
            attn_ops.causal_attention_fwd(b.qkv[l], b.att[l], b.lse[l], b.batch, b.seq, m.n_head)
            self._glin_fwd(b.att[l], p + "attn.c_proj.weight", p + "attn.c_proj.bias", b.tmp_c,
                           g0, Gn, R)
            ln(b.x[l], p + "ln_2", b.ln2[l], st2, delta=b.tmp_c, x_out=b.x_mid[l])
            self._glin_fwd(b.ln2[l], p + "c_fc.weight", p + "c_fc.bias", b.fc[l], g0, Gn, R,
                           gelu_out=b.act[l])
            self._glin_fwd(b.act[l], p + "c_proj.weight", p + "c_proj.bias", b.tmp_c, g0, Gn, R)
            delta = b.tmp_c
        stf = b.stats[2 * L]
        ln(b.x_mid[L - 1], "ln_f", b.lnf, stf, delta=delta, x_out=b.x[L])
        self._glin_fwd(b.lnf, "wte.weight", None, b.logits, g0, Gn, R, n_out=V)
        nn_ops.cross_entropy_fwd_bwd(b.logits, tgt, V, 1.0 / R, b.loss_rows, b.correct_rows)

        # backward
        def ln_b(dy, x, name, st, dres=None, dxsum=None):
            nn_ops.ln_bwd_g(dy, x, wref(name + ".weight"), st[0], st[1], b.dres,
                            gref(name + ".weight"), gref(name + ".bias"), Gn, dres=dres,
                            dxsum=gref(dxsum) if dxsum is not None else None)

        self._glin_bwd(b.logits, b.lnf, "wte.weight", b.dln, g0, Gn, R, n=V)
        ln_b(b.dln, b.x[L], "ln_f", stf, dxsum=f"h.{L - 1}.c_proj.bias")
        for l in reversed(range(L)):
            p = f"h.{l}."
            st1, st2 = b.stats[2 * l], b.stats[2 * l + 1]
            self._glin_bwd(b.dres, b.act[l], p + "c_proj.weight", b.dact, g0, Gn, R)
            nn_ops.gelu_bwd_colsum_g(b.fc[l], b.dact, b.dact, gref(p + "c_fc.bias"), Gn)
            self._glin_bwd(b.dact, b.ln2[l], p + "c_fc.weight", b.dln, g0, Gn, R)
            ln_b(b.dln, b.x_mid[l], p + "ln_2", st2, dres=b.dres, dxsum=p + "attn.c_proj.bias")
            self._glin_bwd(b.dres, b.att[l], p + "attn.c_proj.weight", b.datt, g0, Gn, R)
            attn_ops.causal_attention_bwd(b.qkv[l], b.att[l], b.lse[l], b.datt, b.dqkv,
                                          b.attn_scratch, b.batch, b.seq, m.n_head)
            self._glin_bwd(b.dqkv, b.ln1[l], p + "attn.c_attn.weight", b.dln, g0, Gn, R,
                           also=lambda p=p: nn_ops.colsum_g(b.dqkv, gref(p + "attn.c_attn.bias"), Gn))
            ln_b(b.dln, b.x[l], p + "ln_1", st1, dres=b.dres,
                 dxsum=f"h.{l - 1}.c_proj.bias" if l > 0 else None)
        for j in range(Gn):
            nn_ops.embed_bwd(inp[j * nb : (j + 1) * nb], b.dres[rows[j]], gs[j]["wte.weight"],
                             gs[j]["wpe.weight"], nb, b.seq,
                             sorted_tokens=order[j, 0] if order is not None else None,
                             order=order[j, 1] if order is not None else None)
        n = self.layout.numel
        nn_ops.sgd_step_bf16g(self.pool.data[g0 : g0 + Gn].view(-1), self._ggrads[:Gn].view(-1),
                              self.shadow[g0 : g0 + Gn].view(-1), cfg.learning_rate,
                              cfg.weight_decay)
        assert self.pool.data.stride(0) == n
        self._acc[g0 : g0 + Gn, 0] += b.loss_rows.view(Gn, R).mean(1)
        self._acc[g0 : g0 + Gn, 1] += b.correct_rows.view(Gn, R).mean(1)
        self._acc[g0 : g0 + Gn, 2] += 1.0

    def _train_round_grouped(self, round_idx: int) -> None:
        cfg = self.cfg
        G, B = len(self.pool), cfg.batch_size
        steps = self.shards.steps_per_epoch(0)
        if cfg.graph_capture:
            return self._train_round_graphs(round_idx, steps)
        for epoch in range(cfg.local_epochs):
            self._acc.zero_()
            perms = torch.stack([self.shards.epoch_permutation(k, round_idx, epoch)
                                 for k in range(G)]).view(G, steps, B)
            for s in range(steps):
                for g0 in range(0, G, self.group):
                    Gn = min(self.group, G - g0)
                    rows = perms[g0 : g0 + Gn, s].reshape(-1)
                    self._gstep(g0, Gn, self.shards.inputs[rows])

    def _train_round_graphs(self, round_idx: int, steps: int) -> None:
        """The grouped step as one hipGraph replay per step and group.  What changes from
        step to step is the batch: its tokens and the sort order of its input tokens (for
        the embedding gradient) are assembled on the host for the whole round - the host
        holds a copy of the token shard and draws the shuffles itself - and copied into the
        graph's static input buffers before each replay; the captured step holds no sort,
        no gather and no allocation.  The first step of a (first client, group size) pair
        runs eagerly on the same buffers (it warms the workspaces up), the second is
        captured, every later one is a replay."""
        cfg = self.cfg
        G, B = len(self.pool), cfg.batch_size
        S1 = self.shards.inputs.shape[1]
        if self._host_tokens is None:
            self._host_tokens = self.shards.inputs.cpu()
        groups = [(g0, min(self.group, G - g0)) for g0 in range(0, G, self.group)]
        E = cfg.local_epochs
        # the round's batches, in page-locked memory: [epoch, step, client] blocks
        tok = torch.empty(E, steps, G * B, S1, dtype=torch.int64).pin_memory()
        order = torch.empty(E, steps, G, 2, B * (S1 - 1), dtype=torch.int64).pin_memory()
        for epoch in range(E):
            perms = torch.stack([self.shards.host_permutation(k, round_idx, epoch)
                                 for k in range(G)]).view(G, steps, B)
            for s in range(steps):
                batch = self._host_tokens[perms[:, s].reshape(-1)]  # [G * B, S + 1]
                tok[epoch, s].copy_(batch)
                for k in range(G):
                    srt, idx = nn_ops.token_order(batch[k * B : (k + 1) * B, :-1])
                    order[epoch, s, k, 0].copy_(srt)
                    order[epoch, s, k, 1].copy_(idx)
        for epoch in range(E):
            self._acc.zero_()
            for s in range(steps):
                for g0, Gn in groups:
                    st = self._static_inputs(g0, Gn, S1)
                    st["tokens"].copy_(tok[epoch, s, g0 * B : (g0 + Gn) * B], non_blocking=True)
                    st["order"].copy_(order[epoch, s, g0 : g0 + Gn], non_blocking=True)
                    graph = self._step_graphs.get((g0, Gn))
                    if graph is None and st["warm"]:
                        torch.cuda.synchronize(self.device)
                        graph = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(graph):
                            self._gstep(g0, Gn, st["tokens"], st["order"])
                        self._step_graphs[(g0, Gn)] = graph
                    if graph is not None:
                        graph.replay()
                    else:
                        self._gstep(g0, Gn, st["tokens"], st["order"])
                        st["warm"] = True

    def _static_inputs(self, g0: int, Gn: int, S1: int) -> dict:
        key = (g0, Gn)
        if key not in self._step_inputs:
            B = self.cfg.batch_size
            self._step_inputs[key] = {
                "tokens": torch.zeros(Gn * B, S1, dtype=torch.int64, device=self.device),
                "order": torch.zeros(Gn, 2, B * (S1 - 1), dtype=torch.int64, device=self.device),
                "warm": False,
            }
        return self._step_inputs[key]

    def train_round(self, round_idx: int) -> torch.Tensor:
        cfg = self.cfg
        G = len(self.pool)
        self.metrics.zero_()
        self.shadow.copy_(self.pool.data)  # the round starts from the broadcast model
        if self.group:
            self._train_round_grouped(round_idx)
            return self._round_metrics()
        steps = [self.shards.steps_per_epoch(k) for k in range(G)]
        perms = [
            [self.shards.epoch_permutation(k, round_idx, e) for e in range(cfg.local_epochs)]
            for k in range(G)
        ]
        main = torch.cuda.current_stream() if self._streams else None
        if self._streams:
            for st in self._streams:
                st.wait_stream(main)
        # lane-major groups of clients, step-major inside a group: the lanes'
        # streams always have a step of another client to overlap with
        for first in range(0, G, self.lanes):
            group = list(range(first, min(first + self.lanes, G)))
            for epoch in range(cfg.local_epochs):
                for s in range(max(steps[k] for k in group)):
                    for k in group:
                        if s >= steps[k]:
                            continue
                        lane = k % self.lanes
                        lo = s * cfg.batch_size
                        rows = perms[k][epoch][lo : lo + cfg.batch_size]
                        if self._streams:
                            with torch.cuda.stream(self._streams[lane]):
                                if s == 0:
                                    self._acc[k].zero_()
                                self._step(k, self.shards.inputs[rows], lane)
                        else:
                            if s == 0:
                                self._acc[k].zero_()
                            self._step(k, self.shards.inputs[rows], lane)
        if self._streams:
            for st in self._streams:
                main.wait_stream(st)
        del perms
        return self._round_metrics()

    def _round_metrics(self) -> torch.Tensor:
        G = len(self.pool)
        for k in range(G):
            batches = self._acc[k, 2].clamp(min=1.0)
            self.metrics[k, M_SAMPLES] = float(self.shards.sizes[k])
            self.metrics[k, M_LOSS] = self._acc[k, 0] / batches
            self.metrics[k, M_ACC] = self._acc[k, 1] / batches
            self.metrics[k, M_VALID] = 1.0
        return self.metrics

    @torch.no_grad()
    def evaluate(self, flat: torch.Tensor, batch: int = 4) -> dict[str, float]:
        if self._eval_inputs is None or len(self._eval_inputs) == 0:
            return {}
        self._eval_shadow.copy_(flat)
        w = _views(self.layout, self._eval_shadow)
        xs = self._eval_inputs.to(self.device)
        loss = torch.zeros((), device=self.device)
        hits = torch.zeros((), device=self.device)
        rows = 0
        for lo in range(0, len(xs), batch):
            toks = xs[lo : lo + batch]
            b = self._buffers(toks.shape[0], toks.shape[1] - 1)
            self._forward(w, toks, b)
            loss += b.loss_rows.sum()
            hits += b.correct_rows.sum()
            rows += b.R
        return {"loss": float(loss) / rows, "accuracy": float(hits) / rows,
                "samples": float(len(xs))}


def reference_attention(qkv: torch.Tensor, batch: int, seq: int, heads: int) -> torch.Tensor:
    """``[R, C]`` attention output of a packed ``[R, 3C]`` q|k|v matrix through
    PyTorch (fp32) - the oracle of the attention kernel tests."""
    C = qkv.shape[1] // 3
    q, k, v = qkv.float().view(batch, seq, 3, heads, C // heads).unbind(2)
    y = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2),
                                       is_causal=True)
    return y.transpose(1, 2).reshape(batch * seq, C)
