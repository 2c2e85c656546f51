import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]


def _nb(*sources):
    return {"nbformat": 4, "nbformat_minor": 5, "metadata": {},
            "cells": [{"cell_type": "markdown", "metadata": {}, "source": ["# t"]}] +
                     [{"cell_type": "code", "metadata": {}, "source": [s], "outputs": [],
                       "execution_count": None} for s in sources]}


def test_cells_share_a_namespace_and_outputs_are_stored(tmp_path):
    path = tmp_path / "n.ipynb"
    path.write_text(json.dumps(_nb(
        "import asyncio\nx = 20\nprint('hello')",
        "async def f():\n    await asyncio.sleep(0)\n    return x + 1\ny = await f()\ny * 2",
    )))
    proc = subprocess.run([sys.executable, str(ROOT / "tools" / "execute_notebook.py"), str(path)],
                          capture_output=True, text=True, timeout=120)
    assert proc.returncode == 0, proc.stderr
    cells = [c for c in json.loads(path.read_text())["cells"] if c["cell_type"] == "code"]
    assert cells[0]["outputs"] == [{"output_type": "stream", "name": "stdout", "text": ["hello\n"]}]
    assert cells[1]["execution_count"] == 2
    assert cells[1]["outputs"][0]["data"]["text/plain"] == ["42"]


def test_an_exception_is_recorded_and_ends_the_run(tmp_path):
    path, out = tmp_path / "n.ipynb", tmp_path / "o.ipynb"
    path.write_text(json.dumps(_nb("1/0", "print('never')")))
    proc = subprocess.run([sys.executable, str(ROOT / "tools" / "execute_notebook.py"), str(path),
                           "-o", str(out)], capture_output=True, text=True, timeout=120)
    assert proc.returncode == 1 and "ZeroDivisionError" in proc.stderr
    cells = [c for c in json.loads(out.read_text())["cells"] if c["cell_type"] == "code"]
    assert cells[0]["outputs"][0]["ename"] == "ZeroDivisionError"
    assert cells[1]["outputs"] == [] and cells[1]["execution_count"] is None
