#!/usr/bin/env python
"""Compare the public API surface of the reference with this package, statically.

The reference's sources are only *parsed* (``ast``), never imported or run: for every module
``nanofed/x/y.py`` the tool lists ``__all__``, the public classes with their public methods
and the public functions, each with its parameter names, and then looks the same names up in
``nanofed_amd.x.y``.  Differences are printed; ``--write`` stores the reference's surface as
JSON (``tests/fixtures/reference_api_surface.json``) so that the check keeps running where
the reference's sources are not at hand (``tests/unit/test_api_surface.py``).

    python tools/api_surface_check.py [--reference /root/reference/nanofed] [--write]
"""

from __future__ import annotations

import argparse
import ast
import importlib
import inspect
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FIXTURE = os.path.join(ROOT, "tests", "fixtures", "reference_api_surface.json")


def _params(fn: ast.FunctionDef | ast.AsyncFunctionDef) -> dict:
    a = fn.args
    return {
        "pos": [x.arg for x in a.posonlyargs + a.args],
        "kw": [x.arg for x in a.kwonlyargs],
        "var": a.vararg is not None,
        "varkw": a.kwarg is not None,
    }


def _is_property(fn) -> bool:
    for d in fn.decorator_list:
        if isinstance(d, ast.Name) and d.id in ("property", "cached_property"):
            return True
        if isinstance(d, ast.Attribute) and d.attr in ("setter", "getter", "deleter"):
            return True
    return False


def _is_dataclass(node: ast.ClassDef) -> bool:
    for d in node.decorator_list:
        d = d.func if isinstance(d, ast.Call) else d
        if (isinstance(d, ast.Name) and d.id == "dataclass") or (
            isinstance(d, ast.Attribute) and d.attr == "dataclass"
        ):
            return True
    return False


def _public(name: str) -> bool:
    return not name.startswith("_") or name in ("__init__", "__call__")


def parse_module(path: str) -> dict:
    with open(path, encoding="utf-8") as fh:
        tree = ast.parse(fh.read(), filename=path)
    out: dict = {"all": None, "classes": {}, "functions": {}, "names": []}
    for node in tree.body:
        if isinstance(node, ast.Assign):
            for t in node.targets:
                if isinstance(t, ast.Name):
                    if t.id == "__all__" and isinstance(node.value, (ast.List, ast.Tuple)):
                        out["all"] = [e.value for e in node.value.elts
                                      if isinstance(e, ast.Constant) and isinstance(e.value, str)]
                    elif _public(t.id) and len(t.id) > 1:  # T, P, R: type variables
                        out["names"].append(t.id)
        elif isinstance(node, ast.AnnAssign) and isinstance(node.target, ast.Name):
            if _public(node.target.id):
                out["names"].append(node.target.id)
        elif isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef)) and _public(node.name):
            out["functions"][node.name] = _params(node)
        elif isinstance(node, ast.ClassDef) and _public(node.name):
            methods, fields = {}, []
            for item in node.body:
                if isinstance(item, (ast.FunctionDef, ast.AsyncFunctionDef)) and _public(item.name):
                    methods[item.name] = None if _is_property(item) else _params(item)
                elif isinstance(item, ast.AnnAssign) and isinstance(item.target, ast.Name):
                    if _public(item.target.id):
                        fields.append(item.target.id)
                elif isinstance(item, ast.Assign):
                    for t in item.targets:
                        if isinstance(t, ast.Name) and _public(t.id):
                            fields.append(t.id)
            out["classes"][node.name] = {"methods": methods, "fields": fields,
                                         "dataclass": _is_dataclass(node)}
    return out


def reference_surface(pkg_dir: str) -> dict:
    surface = {}
    for base, _dirs, files in sorted(os.walk(pkg_dir)):
        for f in sorted(files):
            if not f.endswith(".py"):
                continue
            rel = os.path.relpath(os.path.join(base, f), pkg_dir)[:-3].replace(os.sep, ".")
            mod = "" if rel == "__init__" else rel[: -len(".__init__")] if rel.endswith(".__init__") else rel
            surface[mod] = parse_module(os.path.join(base, f))
    return surface


def _own_params(obj) -> dict | None:
    try:
        sig = inspect.signature(obj)
    except (TypeError, ValueError):
        return None
    out: dict = {"pos": [], "kw": [], "var": False, "varkw": False}
    for p in sig.parameters.values():
        if p.kind is p.VAR_POSITIONAL:
            out["var"] = True
        elif p.kind is p.VAR_KEYWORD:
            out["varkw"] = True
        elif p.kind is p.KEYWORD_ONLY:
            out["kw"].append(p.name)
        else:
            out["pos"].append(p.name)
    return out


def _compatible(ref: dict, own: dict | None) -> bool:
    """A call written for the reference's signature binds here: its positional parameters
    come first, in the same order and under the same names, and every keyword is accepted."""
    if own is None:
        return True
    ref_pos = [r for r in ref["pos"] if r not in ("self", "cls")]
    own_pos = [o for o in own["pos"] if o not in ("self", "cls")]
    if own_pos[: len(ref_pos)] != ref_pos and not (own["var"] and own["varkw"]):
        return False
    accepted = set(own_pos) | set(own["kw"])
    return own["varkw"] or all(k in accepted for k in ref["kw"])


def _show(sig: dict | None) -> str:
    if sig is None:
        return "?"
    parts = list(sig["pos"]) + (["*"] if sig["var"] or sig["kw"] else []) + list(sig["kw"])
    return "(" + ", ".join(parts + (["**"] if sig["varkw"] else [])) + ")"


def compare(surface: dict, package: str = "nanofed_amd") -> list[str]:
    problems: list[str] = []
    for mod, info in sorted(surface.items()):
        name = package + ("." + mod if mod else "")
        try:
            module = importlib.import_module(name)
        except Exception as exc:  # noqa: BLE001 - any import failure is the finding
            problems.append(f"{name}: module missing ({type(exc).__name__}: {exc})")
            continue
        for n in info["all"] or []:
            if not hasattr(module, n):
                problems.append(f"{name}: __all__ name {n} missing")
        for n in info["names"]:
            if not hasattr(module, n):
                problems.append(f"{name}: name {n} missing")
        for fn, params in info["functions"].items():
            obj = getattr(module, fn, None)
            if obj is None:
                problems.append(f"{name}: function {fn} missing")
            elif not _compatible(params, _own_params(obj)):
                problems.append(f"{name}.{fn}: {_show(params)} vs {_show(_own_params(obj))}")
        for cn, cinfo in info["classes"].items():
            cls = getattr(module, cn, None)
            if cls is None:
                problems.append(f"{name}: class {cn} missing")
                continue
            for mn, params in cinfo["methods"].items():
                if not hasattr(cls, mn):
                    problems.append(f"{name}.{cn}: method {mn} missing")
                    continue
                if params is None or mn == "__init__" and not inspect.isclass(cls):
                    continue
                own = _own_params(getattr(cls, mn))
                if not _compatible(params, own):
                    problems.append(f"{name}.{cn}.{mn}: {_show(params)} vs {_show(own)}")
            if cinfo.get("dataclass") and "__init__" not in cinfo["methods"]:
                # positional construction: the generated __init__ takes the fields in order
                own = _own_params(cls)
                if own is not None and own["pos"][: len(cinfo["fields"])] != cinfo["fields"]:
                    if not (own["var"] and own["varkw"]):
                        problems.append(f"{name}.{cn}: constructor takes {_show(own)}, the "
                                        f"reference's fields are {cinfo['fields']}")
            annotations = {}
            for klass in getattr(cls, "__mro__", ()):
                annotations.update(getattr(klass, "__annotations__", {}))
            model_fields = getattr(cls, "model_fields", {}) or {}
            for field in cinfo["fields"]:
                if not (hasattr(cls, field) or field in annotations or field in model_fields):
                    problems.append(f"{name}.{cn}: field {field} missing")
    return problems


def main(argv: list[str] | None = None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--reference", default="/root/reference/nanofed")
    ap.add_argument("--write", action="store_true", help="store the parsed surface as the fixture")
    args = ap.parse_args(argv)
    if os.path.isdir(args.reference):
        surface = reference_surface(args.reference)
        if args.write:
            os.makedirs(os.path.dirname(FIXTURE), exist_ok=True)
            with open(FIXTURE, "w") as fh:
                json.dump(surface, fh, indent=1, sort_keys=True)
                fh.write("\n")
    else:
        with open(FIXTURE) as fh:
            surface = json.load(fh)
    problems = compare(surface)
    classes = sum(len(m["classes"]) for m in surface.values())
    functions = sum(len(m["functions"]) for m in surface.values())
    print(f"{len(surface)} modules, {classes} classes, {functions} functions compared; "
          f"{len(problems)} differences")
    for p in problems:
        print("  " + p)
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
