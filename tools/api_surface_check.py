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


_NO_LITERAL = object()


def _literal(node: ast.AST | None):
    """The value of a literal expression (numbers, strings, None, booleans, containers of
    those), or ``_NO_LITERAL``."""
    if node is None:
        return _NO_LITERAL
    if isinstance(node, ast.Call) and getattr(node.func, "id", getattr(node.func, "attr", "")) in (
        "Field", "field"
    ):  # pydantic / dataclasses field declarations: their literal default
        for kw in node.keywords:
            if kw.arg == "default":
                return _literal(kw.value)
        if node.args and getattr(node.func, "id", getattr(node.func, "attr", "")) == "Field":
            return _literal(node.args[0])
        return _NO_LITERAL
    try:
        value = ast.literal_eval(node)
    except (ValueError, TypeError, SyntaxError, MemoryError, RecursionError):
        return _NO_LITERAL
    try:
        return json.loads(json.dumps(value))  # what the fixture can hold (tuples -> lists)
    except (TypeError, ValueError):
        return _NO_LITERAL


def _params(fn: ast.FunctionDef | ast.AsyncFunctionDef) -> dict:
    a = fn.args
    pos = a.posonlyargs + a.args
    defaults: dict = {}
    for arg, d in zip(pos[len(pos) - len(a.defaults):], a.defaults):
        if (v := _literal(d)) is not _NO_LITERAL:
            defaults[arg.arg] = v
    for arg, d in zip(a.kwonlyargs, a.kw_defaults):
        if (v := _literal(d)) is not _NO_LITERAL:
            defaults[arg.arg] = v
    return {
        "pos": [x.arg for x in pos],
        "kw": [x.arg for x in a.kwonlyargs],
        "var": a.vararg is not None,
        "varkw": a.kwarg is not None,
        "defaults": defaults,
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
    out: dict = {"all": None, "classes": {}, "functions": {}, "names": [], "values": {}}
    for node in tree.body:
        if isinstance(node, ast.Assign):
            for t in node.targets:
                if isinstance(t, ast.Name):
                    if t.id == "__all__" and isinstance(node.value, (ast.List, ast.Tuple)):
                        out["all"] = [e.value for e in node.value.elts
                                      if isinstance(e, ast.Constant) and isinstance(e.value, str)]
                    elif _public(t.id) and len(t.id) > 1:  # T, P, R: type variables
                        out["names"].append(t.id)
                        if (v := _literal(node.value)) is not _NO_LITERAL:
                            out["values"][t.id] = v
        elif isinstance(node, ast.AnnAssign) and isinstance(node.target, ast.Name):
            if _public(node.target.id):
                out["names"].append(node.target.id)
                if (v := _literal(node.value)) is not _NO_LITERAL:
                    out["values"][node.target.id] = v
        elif isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef)) and _public(node.name):
            out["functions"][node.name] = _params(node)
        elif isinstance(node, ast.ClassDef) and _public(node.name):
            methods, fields, values = {}, [], {}
            for item in node.body:
                if isinstance(item, (ast.FunctionDef, ast.AsyncFunctionDef)) and _public(item.name):
                    methods[item.name] = None if _is_property(item) else _params(item)
                elif isinstance(item, ast.AnnAssign) and isinstance(item.target, ast.Name):
                    if _public(item.target.id):
                        fields.append(item.target.id)
                        if (v := _literal(item.value)) is not _NO_LITERAL:
                            values[item.target.id] = v
                elif isinstance(item, ast.Assign):
                    for t in item.targets:
                        if isinstance(t, ast.Name) and _public(t.id):
                            fields.append(t.id)
                            if (v := _literal(item.value)) is not _NO_LITERAL:
                                values[t.id] = v
            out["classes"][node.name] = {"methods": methods, "fields": fields, "values": values,
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
    out: dict = {"pos": [], "kw": [], "var": False, "varkw": False, "defaults": {}}
    for p in sig.parameters.values():
        if p.default is not p.empty:
            out["defaults"][p.name] = p.default
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


def _plain(value):
    """A value of this package in the form the fixture holds literals (enums by value)."""
    import enum
    from pathlib import PurePath

    if isinstance(value, enum.Enum):
        value = value.value
    if isinstance(value, PurePath):
        value = str(value)
    try:
        return json.loads(json.dumps(value))
    except (TypeError, ValueError):
        return _NO_LITERAL


def _same(ref, own) -> bool:
    own = _plain(own)
    if own is _NO_LITERAL:
        return True  # not comparable as a literal (objects, factories)
    if isinstance(ref, (int, float)) and isinstance(own, (int, float)) \
            and not isinstance(ref, bool) and not isinstance(own, bool):
        return float(ref) == float(own)
    return ref == own


def _default_problems(where: str, ref: dict, own: dict | None) -> list[str]:
    if own is None:
        return []
    out = []
    for arg, value in ref.get("defaults", {}).items():
        if arg not in own["defaults"]:
            if arg in own["pos"] or arg in own["kw"]:
                out.append(f"{where}: {arg} has default {value!r} in the reference, none here")
        elif not _same(value, own["defaults"][arg]):
            out.append(f"{where}: default of {arg} is {own['defaults'][arg]!r}, "
                       f"the reference's {value!r}")
    return out


def _field_default(cls, field: str):
    import dataclasses

    model_fields = getattr(cls, "model_fields", None) or {}
    if field in model_fields:
        info = model_fields[field]
        return _NO_LITERAL if info.default_factory is not None else info.default
    if dataclasses.is_dataclass(cls):
        for f in dataclasses.fields(cls):
            if f.name == field:
                return _NO_LITERAL if f.default is dataclasses.MISSING else f.default
    return getattr(cls, field, _NO_LITERAL)


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
        for n, value in info.get("values", {}).items():
            if hasattr(module, n) and not _same(value, getattr(module, n)):
                problems.append(f"{name}.{n} is {getattr(module, n)!r}, the reference's {value!r}")
        for fn, params in info["functions"].items():
            obj = getattr(module, fn, None)
            if obj is None:
                problems.append(f"{name}: function {fn} missing")
            elif not _compatible(params, _own_params(obj)):
                problems.append(f"{name}.{fn}: {_show(params)} vs {_show(_own_params(obj))}")
            else:
                problems += _default_problems(f"{name}.{fn}", params, _own_params(obj))
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
                else:
                    problems += _default_problems(f"{name}.{cn}.{mn}", params, own)
            if cinfo.get("dataclass") and "__init__" not in cinfo["methods"]:
                # positional construction: the generated __init__ takes the fields in order
                own = _own_params(cls)
                if own is not None and own["pos"][: len(cinfo["fields"])] != cinfo["fields"]:
                    if not (own["var"] and own["varkw"]):
                        problems.append(f"{name}.{cn}: constructor takes {_show(own)}, the "
                                        f"reference's fields are {cinfo['fields']}")
            for field, value in cinfo.get("values", {}).items():
                own_value = _field_default(cls, field)
                if own_value is not _NO_LITERAL and not _same(value, own_value):
                    problems.append(f"{name}.{cn}.{field} defaults to {own_value!r}, "
                                    f"the reference's {value!r}")
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
