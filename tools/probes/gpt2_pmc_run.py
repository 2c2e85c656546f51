"""A few eager (not graph-captured) steps of the grouped GPT-2 trainer at the benchmark
shape (8 clients x 4 sequences of 1024 tokens), as the program of tools/pmc_passes.sh."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nanofed_amd.engine import EngineConfig, FederatedEngine  # noqa: E402
from nanofed_amd.parallel import Communicator  # noqa: E402

torch.cuda.set_device(0)
cfg = EngineConfig(model="gpt2_125m", dataset="synthetic_tokens", num_clients=8,
                   samples_per_client=8, batch_size=4, local_epochs=1, seq_len=1024,
                   learning_rate=0.2, eval_samples=8, token_active_vocab=2048, graph_capture=False)
eng = FederatedEngine(cfg, comm=Communicator(device=torch.device("cuda", 0)))
rep = eng.run_round()
torch.cuda.synchronize()
print(type(eng.trainer).__name__, rep.status, rep.agg_metrics)
