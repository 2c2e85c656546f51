"""Federated GPT-2 (124M by default) on token sequences (synthetic by default).

    python examples/gpt2/run_federated.py --clients 8 --rounds 2
    python examples/gpt2/run_federated.py --layers 4 --embd 256 --heads 4 --vocab 8192 --seq-len 256
    python examples/gpt2/run_federated.py --data-dir ~/data/openwebtext   # train.bin / val.bin

On a GPU the native trainer runs the step by hand, all clients of the GPU in every launch:
hand-written bf16 MFMA GEMMs (batched, per-client weights), flash attention forward /
backward, residual+LayerNorm, GELU, fused cross-entropy, embedding.  The 498 MB fp32 arena
of the 124M model is reduced with reduce-scatter + all-gather when several GPUs take part,
so all xGMI links carry a share of it.  ``--data-dir`` names a folder with ``train.bin`` and
``val.bin``: flat arrays of token ids (``uint16`` unless ``--token-dtype`` says otherwise;
``.npy`` works too), memory-mapped and cut into windows of ``seq_len + 1`` tokens.
"""

from __future__ import annotations

import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from nanofed_amd.engine import EngineConfig, FederatedEngine  # noqa: E402
from nanofed_amd.parallel import Communicator  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--clients", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--sequences-per-client", type=int, default=16)
    ap.add_argument("--batch-size", type=int, default=4)
    ap.add_argument("--seq-len", type=int, default=1024)
    ap.add_argument("--lr", type=float, default=0.01)
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--embd", type=int, default=768)
    ap.add_argument("--heads", type=int, default=12)
    ap.add_argument("--vocab", type=int, default=50257)
    ap.add_argument("--data-dir", default=None, help="folder with train.bin / val.bin (token ids)")
    ap.add_argument("--token-dtype", default="uint16", help="element type of the .bin files")
    args = ap.parse_args()
    comm = Communicator.from_env()
    kwargs = {}
    if (args.layers, args.embd, args.heads, args.vocab, args.seq_len) != (12, 768, 12, 50257, 1024):
        kwargs = dict(n_layer=args.layers, n_embd=args.embd, n_head=args.heads,
                      vocab_size=args.vocab, n_positions=args.seq_len)
    cfg = EngineConfig(
        model="gpt2_125m", dataset="tokens" if args.data_dir else "synthetic_tokens",
        data_dir=os.path.expanduser(args.data_dir or "./data"), token_dtype=args.token_dtype,
        num_clients=args.clients,
        samples_per_client=args.sequences_per_client, batch_size=args.batch_size,
        local_epochs=1, seq_len=args.seq_len, learning_rate=args.lr, eval_samples=8,
        model_kwargs=kwargs,
    )
    engine = FederatedEngine(cfg, comm=comm)
    if comm.is_root:
        print(json.dumps({"trainer": type(engine.trainer).__name__,
                          "arena_mb": round(engine.layout.numel * 4 / 1e6, 1)}))
    for report in engine.run(args.rounds):
        if comm.is_root:
            print(json.dumps({"round": report.round_id, "status": report.status,
                              **report.agg_metrics, "wall_s": round(report.wall_s, 3)}))
    if comm.is_root:
        print(json.dumps({"eval": engine.evaluate()}))
    engine.close()
    comm.finish()  # the process group is destroyed before the interpreter shuts down


if __name__ == "__main__":
    main()
