#!/usr/bin/env python3
"""Generate CHANGELOG.md (and a release-notes page) from conventional commits.

    python scripts/changelog.py [--since TAG] [--version X.Y.Z]

Commits are grouped by type (`feat`, `fix`, `perf`, `docs`, `test`, `refactor`,
`build`, `ci`, `chore`); subjects without a type prefix go under "Other".
"""

from __future__ import annotations

import argparse
import re
import subprocess
from collections import OrderedDict
from datetime import date
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SECTIONS = OrderedDict([
    ("feat", "Features"), ("fix", "Bug fixes"), ("perf", "Performance"), ("docs", "Documentation"),
    ("test", "Tests"), ("refactor", "Refactoring"), ("build", "Build"), ("ci", "CI"),
    ("chore", "Chores"), ("other", "Other"),
])
PATTERN = re.compile(r"^(?P<type>[a-z]+)(\((?P<scope>[^)]+)\))?(?P<bang>!)?:\s*(?P<subject>.+)$")


def git(*args: str) -> str:
    return subprocess.run(["git", *args], cwd=ROOT, check=True, capture_output=True,
                          text=True).stdout.strip()


def commits(since: str | None) -> list[tuple[str, str]]:
    rng = f"{since}..HEAD" if since else "HEAD"
    out = git("log", rng, "--pretty=format:%h%x09%s")
    return [tuple(line.split("\t", 1)) for line in out.splitlines() if "\t" in line]


def group(entries: list[tuple[str, str]]) -> dict[str, list[str]]:
    buckets: dict[str, list[str]] = {k: [] for k in SECTIONS}
    for sha, subject in entries:
        m = PATTERN.match(subject)
        kind = m.group("type") if m and m.group("type") in SECTIONS else "other"
        text = m.group("subject") if m and kind != "other" else subject
        scope = f"**{m.group('scope')}**: " if m and m.group("scope") else ""
        breaking = " (BREAKING)" if m and m.group("bang") else ""
        buckets[kind].append(f"- {scope}{text}{breaking} ({sha})")
    return buckets


def render(version: str, buckets: dict[str, list[str]]) -> str:
    lines = [f"## {version} - {date.today().isoformat()}", ""]
    for key, title in SECTIONS.items():
        if buckets[key]:
            lines += [f"### {title}", "", *buckets[key], ""]
    return "\n".join(lines)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--since")
    ap.add_argument("--version", default="Unreleased")
    args = ap.parse_args()
    since = args.since
    if since is None:
        try:
            since = git("describe", "--tags", "--abbrev=0")
        except subprocess.CalledProcessError:
            since = None
    body = render(args.version, group(commits(since)))
    path = ROOT / "CHANGELOG.md"
    old = path.read_text() if path.exists() else "# Changelog\n\n"
    head, _, rest = old.partition("\n\n")
    path.write_text(f"{head}\n\n{body}\n{rest}")
    notes = ROOT / "docs" / "release_notes"
    notes.mkdir(parents=True, exist_ok=True)
    (notes / f"{args.version}.md").write_text(body)
    (notes / "latest.md").write_text(body)
    print(body)


if __name__ == "__main__":
    main()
