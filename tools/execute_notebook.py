"""Execute a Jupyter notebook without Jupyter and store the outputs in it.

    python tools/execute_notebook.py examples/mnist/tutorial.ipynb [-o executed.ipynb]

Code cells run in order in one namespace, like a kernel would run them; top-level
``await`` works (cells are compiled with ``PyCF_ALLOW_TOP_LEVEL_AWAIT`` and driven
by one event loop).  What a cell prints becomes a ``stream`` output, the value of a
trailing expression an ``execute_result``, an exception an ``error`` output (and
ends the run with status 1).  Only the nbformat-4 JSON layout is needed for that,
so the tool has no dependency beyond the standard library.
"""

from __future__ import annotations

import argparse
import ast
import asyncio
import contextlib
import inspect
import io
import json
import os
import sys
import traceback


def _split_lines(text: str) -> list[str]:
    return text.splitlines(keepends=True)


def run_cell(source: str, namespace: dict, loop: asyncio.AbstractEventLoop, number: int) -> list[dict]:
    outputs: list[dict] = []
    stdout, stderr = io.StringIO(), io.StringIO()
    tree = ast.parse(source, filename=f"<cell {number}>", mode="exec")
    last_expr = None
    if tree.body and isinstance(tree.body[-1], ast.Expr):
        last_expr = ast.Expression(tree.body.pop().value)
    flags = ast.PyCF_ALLOW_TOP_LEVEL_AWAIT
    error: dict | None = None
    value = None
    with contextlib.redirect_stdout(stdout), contextlib.redirect_stderr(stderr):
        try:
            for node, mode in ((tree, "exec"), (last_expr, "eval")):
                if node is None:
                    continue
                code = compile(node, f"<cell {number}>", mode, flags=flags)
                result = eval(code, namespace)  # noqa: S307 - that is what a kernel does
                if inspect.iscoroutine(result):
                    result = loop.run_until_complete(result)
                if mode == "eval":
                    value = result
        except BaseException as exc:  # noqa: BLE001 - recorded in the notebook
            error = {
                "output_type": "error", "ename": type(exc).__name__, "evalue": str(exc),
                "traceback": traceback.format_exc().splitlines(),
            }
    if stdout.getvalue():
        outputs.append({"output_type": "stream", "name": "stdout", "text": _split_lines(stdout.getvalue())})
    if stderr.getvalue():
        outputs.append({"output_type": "stream", "name": "stderr", "text": _split_lines(stderr.getvalue())})
    if value is not None:
        outputs.append({
            "output_type": "execute_result", "execution_count": number, "metadata": {},
            "data": {"text/plain": _split_lines(repr(value))},
        })
    if error is not None:
        outputs.append(error)
    return outputs


def execute(notebook: dict) -> bool:
    """Run every code cell of ``notebook`` in place; False when a cell raised."""
    namespace: dict = {"__name__": "__main__"}
    if os.getcwd() not in sys.path:
        sys.path.insert(0, os.getcwd())  # a kernel imports from its working directory
    loop = asyncio.new_event_loop()
    asyncio.set_event_loop(loop)
    count = 0
    ok = True
    try:
        for cell in notebook["cells"]:
            if cell.get("cell_type") != "code":
                continue
            count += 1
            source = "".join(cell.get("source", []))
            print(f"[cell {count}] {source.splitlines()[0] if source.strip() else ''}", file=sys.stderr, flush=True)
            cell["outputs"] = run_cell(source, namespace, loop, count) if ok else []
            cell["execution_count"] = count if ok else None
            if any(o["output_type"] == "error" for o in cell["outputs"]):
                ok = False
    finally:
        loop.close()
    return ok


def main() -> None:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("notebook")
    ap.add_argument("-o", "--output", help="where to write the executed notebook (default: in place)")
    args = ap.parse_args()
    with open(args.notebook) as f:
        notebook = json.load(f)
    if notebook.get("nbformat") != 4:
        raise SystemExit("only nbformat 4 notebooks are supported")
    ok = execute(notebook)
    with open(args.output or args.notebook, "w") as f:
        json.dump(notebook, f, indent=1, ensure_ascii=False)
        f.write("\n")
    if not ok:
        for cell in notebook["cells"]:
            for out in cell.get("outputs", []):
                if out["output_type"] == "error":
                    print("\n".join(out["traceback"]), file=sys.stderr)
        raise SystemExit(1)


if __name__ == "__main__":
    main()
