#!/usr/bin/env python3
"""Release gate: semantic version check, clean tree, native build, CPU tests,
then bump the version in pyproject.toml / nanofed_amd/__init__.py and write
the release notes.

    python scripts/prepare_release.py 0.2.0 [--skip-tests]
"""

from __future__ import annotations

import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SEMVER = re.compile(r"^\d+\.\d+\.\d+(?:-(?:alpha|beta|rc)\.\d+)?$")


def run(*cmd: str) -> None:
    print("+", " ".join(cmd))
    subprocess.run(cmd, cwd=ROOT, check=True)


def main() -> None:
    if len(sys.argv) < 2 or not SEMVER.match(sys.argv[1]):
        raise SystemExit("usage: prepare_release.py X.Y.Z[-rc.N] [--skip-tests]")
    version = sys.argv[1]
    dirty = subprocess.run(["git", "status", "--porcelain"], cwd=ROOT, capture_output=True,
                           text=True).stdout.strip()
    if dirty:
        raise SystemExit("working tree is not clean:\n" + dirty)
    run(sys.executable, "-m", "nanofed_amd.native_build")
    if "--skip-tests" not in sys.argv:
        run(sys.executable, "-m", "pytest", "tests", "-q", "-m", "not gpu")
    for path, pattern, repl in (
        (ROOT / "pyproject.toml", r'(?m)^version = "[^"]+"', f'version = "{version}"'),
        (ROOT / "nanofed_amd" / "__init__.py", r'__version__ = "[^"]+"', f'__version__ = "{version}"'),
    ):
        path.write_text(re.sub(pattern, repl, path.read_text(), count=1))
    run(sys.executable, "scripts/changelog.py", "--version", version)
    print(f"prepared {version}: review the diff, then `make release VERSION={version}`")


if __name__ == "__main__":
    main()
