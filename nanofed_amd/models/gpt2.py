"""GPT-2 small (124M parameters: 12 layers, 12 heads, d=768, vocab 50257,
context 1024) as a causal language model.

Used for the large-arena federated configuration, where the cost of a round
is dominated by the size of the parameter reduce.  Attention goes through
``scaled_dot_product_attention`` (flash kernel on ROCm); the token embedding
and the LM head share one weight, so the arena holds it once.
"""

from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass(frozen=True)
class GPT2Config:
    vocab_size: int = 50257
    n_positions: int = 1024
    n_embd: int = 768
    n_layer: int = 12
    n_head: int = 12
    dropout: float = 0.0


class _Attention(nn.Module):
    def __init__(self, cfg: GPT2Config) -> None:
        super().__init__()
        self.n_head = cfg.n_head
        self.c_attn = nn.Linear(cfg.n_embd, 3 * cfg.n_embd)
        self.c_proj = nn.Linear(cfg.n_embd, cfg.n_embd)
        self.dropout = cfg.dropout

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        b, t, c = x.shape
        q, k, v = self.c_attn(x).split(c, dim=2)
        hd = c // self.n_head
        q = q.view(b, t, self.n_head, hd).transpose(1, 2)
        k = k.view(b, t, self.n_head, hd).transpose(1, 2)
        v = v.view(b, t, self.n_head, hd).transpose(1, 2)
        y = F.scaled_dot_product_attention(
            q, k, v, is_causal=True,
            dropout_p=self.dropout if self.training else 0.0,
        )
        return self.c_proj(y.transpose(1, 2).reshape(b, t, c))


class _Block(nn.Module):
    def __init__(self, cfg: GPT2Config) -> None:
        super().__init__()
        self.ln_1 = nn.LayerNorm(cfg.n_embd)
        self.attn = _Attention(cfg)
        self.ln_2 = nn.LayerNorm(cfg.n_embd)
        self.c_fc = nn.Linear(cfg.n_embd, 4 * cfg.n_embd)
        self.c_proj = nn.Linear(4 * cfg.n_embd, cfg.n_embd)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = x + self.attn(self.ln_1(x))
        h = F.gelu(self.c_fc(self.ln_2(x)), approximate="tanh")
        return x + self.c_proj(h)


class GPT2LM(nn.Module):
    """Returns next-token logits ``[B, T, vocab]``; ``loss(tokens)`` gives the
    mean next-token cross-entropy."""

    def __init__(self, cfg: GPT2Config | None = None) -> None:
        super().__init__()
        self.cfg = cfg or GPT2Config()
        c = self.cfg
        self.wte = nn.Embedding(c.vocab_size, c.n_embd)
        self.wpe = nn.Embedding(c.n_positions, c.n_embd)
        self.h = nn.ModuleList(_Block(c) for _ in range(c.n_layer))
        self.ln_f = nn.LayerNorm(c.n_embd)
        self.apply(self._init)
        for name, p in self.named_parameters():
            if name.endswith("c_proj.weight"):
                nn.init.normal_(p, 0.0, 0.02 / math.sqrt(2 * c.n_layer))

    @staticmethod
    def _init(m: nn.Module) -> None:
        if isinstance(m, (nn.Linear, nn.Embedding)):
            nn.init.normal_(m.weight, 0.0, 0.02)
        if isinstance(m, nn.Linear) and m.bias is not None:
            nn.init.zeros_(m.bias)

    def forward(self, idx: torch.Tensor) -> torch.Tensor:
        _, t = idx.shape
        pos = torch.arange(t, device=idx.device)
        x = self.wte(idx) + self.wpe(pos)
        for block in self.h:
            x = block(x)
        return F.linear(self.ln_f(x), self.wte.weight)

    def loss(self, tokens: torch.Tensor) -> torch.Tensor:
        logits = self.forward(tokens[:, :-1])
        return F.cross_entropy(
            logits.reshape(-1, logits.size(-1)).float(),
            tokens[:, 1:].reshape(-1),
        )
