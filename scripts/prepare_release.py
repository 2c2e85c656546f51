#!/usr/bin/env python3
"""Release gate.

    python scripts/prepare_release.py 0.2.0 [--skip tests,types] [--allow-dirty]

Checks, in this order, and stops at the summary if any of them failed:

1. the version is semantic (``X.Y.Z`` or ``X.Y.Z-rc.N``) and greater than the current one;
2. the working tree is clean;
3. the native libraries build from source (host C++ and gfx950 HIP);
4. the CPU test suite passes;
5. the linter passes (``ruff check`` when installed, else a byte-compile of every source);
6. the type checker passes (``mypy`` with the configuration in ``pyproject.toml``);
7. the package builds (``pip wheel``).

A tool that is not installed is reported as *unavailable*, which fails the gate
unless the check is named in ``--skip``.  Then the version is written to
``pyproject.toml`` and ``nanofed_amd/__init__.py`` and the release-notes page
``docs/source/release_notes/vX.Y.Z.rst`` is created from ``template.rst``.

Reference: scripts/prepare_release.py:9-139 (same gates: clean tree, tests, mypy,
ruff, build; same release-notes step).
"""

from __future__ import annotations

import argparse
import importlib.util
import re
import shutil
import subprocess
import sys
import tempfile
from datetime import datetime, timezone
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SEMVER = re.compile(r"^(\d+)\.(\d+)\.(\d+)(?:-(alpha|beta|rc)\.(\d+))?$")
NOTES_DIR = ROOT / "docs" / "source" / "release_notes"
LINT_TARGETS = ["nanofed_amd", "scripts", "tests", "tools", "bench.py"]


def version_key(version: str) -> tuple:
    m = SEMVER.match(version)
    if not m:
        raise ValueError(f"{version!r} is not a semantic version (X.Y.Z or X.Y.Z-rc.N)")
    stage = {"alpha": 0, "beta": 1, "rc": 2, None: 3}[m.group(4)]
    return (int(m.group(1)), int(m.group(2)), int(m.group(3)), stage, int(m.group(5) or 0))


def current_version() -> str:
    text = (ROOT / "pyproject.toml").read_text()
    m = re.search(r'(?m)^version\s*=\s*"([^"]+)"', text)
    if not m:
        raise RuntimeError("pyproject.toml has no version")
    return m.group(1)


def write_version(version: str) -> None:
    for path, pattern, repl in (
        (ROOT / "pyproject.toml", r'(?m)^version\s*=\s*"[^"]+"', f'version = "{version}"'),
        (ROOT / "nanofed_amd" / "__init__.py", r'__version__\s*=\s*"[^"]+"', f'__version__ = "{version}"'),
    ):
        path.write_text(re.sub(pattern, repl, path.read_text(), count=1))


def create_release_notes(version: str) -> Path | None:
    page = NOTES_DIR / f"v{version}.rst"
    if page.exists():
        return page
    template = NOTES_DIR / "template.rst"
    if not template.exists():
        print(f"no template at {template.relative_to(ROOT)}: release notes not created")
        return None
    today = datetime.now(timezone.utc).strftime("%B %d, %Y")
    page.write_text(template.read_text().replace("{version}", version).replace("{date}", today))
    index = NOTES_DIR / "index.rst"
    if index.exists() and f"v{version}" not in index.read_text():
        lines = index.read_text().rstrip("\n").splitlines()
        at = next((i for i, ln in enumerate(lines) if ln.strip().startswith("v")), len(lines))
        lines.insert(at, f"   v{version}")
        index.write_text("\n".join(lines) + "\n")
    return page


def _run(cmd: list[str]) -> bool:
    print("+", " ".join(cmd), flush=True)
    return subprocess.run(cmd, cwd=ROOT).returncode == 0


def check_clean() -> str:
    out = subprocess.run(["git", "status", "--porcelain"], cwd=ROOT, capture_output=True, text=True)
    return "ok" if out.returncode == 0 and not out.stdout.strip() else "failed"


def check_build() -> str:
    return "ok" if _run([sys.executable, "-m", "nanofed_amd.native_build"]) else "failed"


def check_tests() -> str:
    return "ok" if _run([sys.executable, "-m", "pytest", "tests", "-q", "-m", "not gpu"]) else "failed"


def check_lint() -> str:
    if shutil.which("ruff"):
        return "ok" if _run(["ruff", "check", *LINT_TARGETS]) else "failed"
    # without ruff: every source must at least compile
    ok = _run([sys.executable, "-m", "compileall", "-q", *LINT_TARGETS])
    return "ok (byte-compile only: ruff is not installed)" if ok else "failed"


def check_types() -> str:
    if importlib.util.find_spec("mypy") is None:
        return "unavailable (mypy is not installed)"
    return "ok" if _run([sys.executable, "-m", "mypy", "nanofed_amd"]) else "failed"


def check_package() -> str:
    with tempfile.TemporaryDirectory() as out:
        ok = _run([sys.executable, "-m", "pip", "wheel", ".", "--no-deps", "--no-build-isolation",
                   "-w", out, "-q"])
        return "ok" if ok and any(Path(out).glob("*.whl")) else "failed"


CHECKS = [
    ("clean", "working tree is clean", check_clean),
    ("build", "native libraries build", check_build),
    ("tests", "CPU tests pass", check_tests),
    ("lint", "lint passes", check_lint),
    ("types", "type check passes", check_types),
    ("package", "wheel builds", check_package),
]


def run_checks(skip: set[str]) -> tuple[bool, list[tuple[str, str]]]:
    results, ok = [], True
    for key, title, fn in CHECKS:
        if key in skip:
            results.append((title, "skipped"))
            continue
        verdict = fn()
        results.append((title, verdict))
        ok = ok and verdict.startswith("ok")
    return ok, results


def main(argv: list[str] | None = None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("new_version")
    ap.add_argument("--skip", default="", help="comma-separated checks to skip: "
                    + ", ".join(k for k, _, _ in CHECKS))
    ap.add_argument("--allow-dirty", action="store_true", help="same as --skip clean")
    args = ap.parse_args(argv)
    skip = {s for s in args.skip.split(",") if s}
    if args.allow_dirty:
        skip.add("clean")
    unknown = skip - {k for k, _, _ in CHECKS}
    if unknown:
        print(f"unknown checks in --skip: {sorted(unknown)}", file=sys.stderr)
        return 2
    try:
        new, old = version_key(args.new_version), version_key(current_version())
    except ValueError as exc:
        print(exc, file=sys.stderr)
        return 2
    if new <= old:
        print(f"{args.new_version} is not greater than the current version {current_version()}", file=sys.stderr)
        return 2
    print(f"current version {current_version()} -> preparing {args.new_version}")
    ok, results = run_checks(skip)
    print("\nPre-release checks")
    for title, verdict in results:
        print(f"  [{'x' if verdict.startswith('ok') else '-' if verdict == 'skipped' else ' '}] {title}: {verdict}")
    if not ok:
        print("\nNot ready: fix the checks above (or name them in --skip) and run again.")
        return 1
    write_version(args.new_version)
    page = create_release_notes(args.new_version)
    print(f"\nVersion set to {args.new_version}.")
    if page is not None:
        print(f"Release notes: {page.relative_to(ROOT)} - fill in the overview, then `make release`.")
    return 0


if __name__ == "__main__":
    sys.exit(main())
