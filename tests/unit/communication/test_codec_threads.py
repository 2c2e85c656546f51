"""The JSON tensor codec's parallel paths and their switches: the text of a large tensor is
the same with eight threads, with NANOFED_JSON_THREADS=1 (parallel paths off) and with
the parallel decoder's size limit at zero, and it decodes to the same values either way."""

import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

SCRIPT = r"""
import hashlib, sys, torch
sys.path.insert(0, {root!r})
from nanofed_amd.communication.codec import decode_state_json, encode_state_json, native_available
assert native_available()
torch.manual_seed(0)
state = {{"w": torch.randn(700, 600), "b": torch.randn(600)}}   # 420,000 numbers, ~5 MB of text
text = encode_state_json(state)
back = decode_state_json(text)
assert torch.equal(back["w"], state["w"]) and torch.equal(back["b"], state["b"])
print(len(text), hashlib.sha256(text).hexdigest())
"""


def _run(**env):
    proc = subprocess.run([sys.executable, "-c", SCRIPT.format(root=ROOT)], capture_output=True,
                          text=True, timeout=300, env=dict(os.environ, **env))
    assert proc.returncode == 0, proc.stderr[-2000:]
    return proc.stdout.strip()


def test_same_text_with_and_without_worker_threads():
    many = _run(NANOFED_JSON_THREADS="8")
    one = _run(NANOFED_JSON_THREADS="1")
    capped = _run(NANOFED_JSON_THREADS="8", NANOFED_JSON_PARALLEL_MAX_MB="0")
    assert many == one == capped
    assert int(many.split()[0]) > (1 << 21)  # long enough for the parallel decoder
