"""Three-rank gloo jobs that end the ordinary way (no os._exit): how often does a rank abort, and where?"""
import os, subprocess, sys, time, socket, tempfile, pathlib
ROOT = "/root/repo"
WORKER = r'''
import sys, os, faulthandler, torch
faulthandler.enable(all_threads=True)
sys.path.insert(0, "/root/repo")
from nanofed_amd.engine import EngineConfig, FederatedEngine
from nanofed_amd.parallel import Communicator
comm = Communicator.from_env(backend="gloo", device="cpu")
cfg = EngineConfig(model="mnist_mlp", num_clients=8, samples_per_client=[96, 64, 64, 32, 64, 64, 96, 32],
                   batch_size=32, local_epochs=1, eval_samples=64, dtype="fp32", backend="torch",
                   model_kwargs={"hidden": 16}, learning_rate=0.1)
eng = FederatedEngine(cfg, comm=comm, base_dir=os.environ.get("BASE") if comm.is_root else None)
eng.run(2)
if os.environ.get("NO_ENGINE_CLOSE") != "1":
    eng.close()
mode = os.environ.get("EXIT_MODE", "plain")
if mode == "finish":
    comm.finish(0)
elif mode == "close":
    comm.close()
elif mode == "guard":
    pass  # the atexit guard of from_env closes the group
'''
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); return s.getsockname()[1]
def run(mode, i, tmp):
    script = tmp / "w.py"; script.write_text(WORKER)
    port = free_port()
    procs = []
    for r in range(3):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="3", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1", EXIT_MODE=mode, BASE=str(tmp / f"run{i}"))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    out = []
    for p in procs:
        try:
            p.wait(timeout=120)
        except subprocess.TimeoutExpired:
            p.kill()
        out.append((p.returncode, p.stderr.read()))
    return out
if __name__ == "__main__":
    mode, n = sys.argv[1], int(sys.argv[2])
    tmp = pathlib.Path(tempfile.mkdtemp())
    bad = 0
    for i in range(n):
        res = run(mode, i, tmp)
        codes = [c for c, _ in res]
        if codes != [0, 0, 0]:
            bad += 1
            print(f"job {i}: codes {codes}")
            for c, e in res:
                if c != 0:
                    print("---- stderr tail:\n" + e[-2500:])
            sys.stdout.flush()
    print(f"mode {mode}: {bad} bad of {n}")
