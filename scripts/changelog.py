#!/usr/bin/env python3
"""Changelog and release notes from conventional commits.

    python scripts/changelog.py 0.2.0            # since the latest tag
    python scripts/changelog.py 0.2.0 --since v0.1.0 --dry-run
    python scripts/changelog.py 0.2.0 --replace  # again, after more commits

Reads ``git log`` since the latest tag (or ``--since``), sorts the subjects by their
conventional-commit type (``feat(scope): subject``) and writes

* a section ``## [X.Y.Z] - date`` at the top of ``CHANGELOG.md`` (grouped by type,
  with links to the commits when the repository has a GitHub remote); a version
  that already has a section is left alone;
* the list after the marker line of ``docs/source/release_notes/vX.Y.Z.rst`` (the
  page ``scripts/prepare_release.py`` creates from ``template.rst``), replacing
  whatever an earlier run put there.

Reference: scripts/changelog.py:8-282 (same inputs and outputs).
"""

from __future__ import annotations

import argparse
import re
import subprocess
import sys
from collections import OrderedDict
from dataclasses import dataclass
from datetime import datetime, timezone
from pathlib import Path

SECTIONS = OrderedDict([
    ("feat", "Features"), ("fix", "Bug Fixes"), ("perf", "Performance"),
    ("docs", "Documentation"), ("style", "Style"), ("refactor", "Code Refactoring"),
    ("test", "Tests"), ("build", "Build System"), ("ci", "Continuous Integration"),
    ("chore", "Chores"), ("other", "Other Changes"),
])
SUBJECT = re.compile(r"^(?P<type>[a-zA-Z]+)(?:\((?P<scope>[^)]+)\))?(?P<bang>!)?:\s+(?P<text>.+)$")
RST_MARKER = ".. Generated automatically from git commits"
HEADER = "# Changelog"


@dataclass(frozen=True)
class Commit:
    sha: str
    kind: str
    scope: str
    text: str
    breaking: bool


class Repo:
    def __init__(self, root: str | Path = ".") -> None:
        self.root = Path(root).resolve()
        if self._git("rev-parse", "--git-dir", check=False) is None:
            raise RuntimeError(f"{self.root} is not inside a git repository")

    def _git(self, *args: str, check: bool = True) -> str | None:
        proc = subprocess.run(["git", *args], cwd=self.root, capture_output=True, text=True)
        if proc.returncode != 0:
            if check:
                raise RuntimeError(f"git {' '.join(args)}: {proc.stderr.strip()}")
            return None
        return proc.stdout.strip()

    def web_url(self) -> str:
        """``https://github.com/owner/name`` of the ``origin`` remote, or ''."""
        url = self._git("config", "--get", "remote.origin.url", check=False) or ""
        m = re.match(r"^(?:git@|ssh://git@)([^:/]+)[:/](.+?)(?:\.git)?$", url)
        if m:
            return f"https://{m.group(1)}/{m.group(2)}"
        if url.startswith("http"):
            return url[:-4] if url.endswith(".git") else url
        return ""

    def latest_tag(self) -> str | None:
        return self._git("describe", "--tags", "--abbrev=0", check=False) or None

    def commits(self, since: str | None) -> list[Commit]:
        rng = [f"{since}..HEAD"] if since else []
        out = self._git("log", *rng, "--no-merges", "--pretty=format:%h%x09%s") or ""
        return [parse(sha, subject) for sha, _, subject in
                (line.partition("\t") for line in out.splitlines()) if subject]


_ALIASES = {"tests": "test", "feature": "feat", "bugfix": "fix", "doc": "docs"}


def parse(sha: str, subject: str) -> Commit:
    m = SUBJECT.match(subject)
    if m:
        kind = m.group("type").lower()
        kind = _ALIASES.get(kind, kind)
        if kind in SECTIONS and kind != "other":
            return Commit(sha, kind, m.group("scope") or "", m.group("text"), bool(m.group("bang")))
    return Commit(sha, "other", "", subject, False)


def group(commits: list[Commit]) -> "OrderedDict[str, list[Commit]]":
    buckets: OrderedDict[str, list[Commit]] = OrderedDict((k, []) for k in SECTIONS)
    for c in commits:
        buckets[c.kind].append(c)
    return buckets


def _line(c: Commit) -> str:
    scope = f"({c.scope}) " if c.scope else ""
    return f"{scope}{c.text}{' [BREAKING]' if c.breaking else ''}"


def render_markdown(version: str, buckets: dict[str, list[Commit]], url: str, day: str) -> str:
    v = version.lstrip("v")
    title = f"## [{v}]({url}/releases/tag/v{v}) - {day}" if url else f"## [{v}] - {day}"
    lines = [title, ""]
    for kind, heading in SECTIONS.items():
        if not buckets.get(kind):
            continue
        lines += [f"### {heading}", ""]
        for c in buckets[kind]:
            ref = f"([{c.sha}]({url}/commit/{c.sha}))" if url else f"({c.sha})"
            lines.append(f"- {_line(c)} {ref}")
        lines.append("")
    return "\n".join(lines)


def render_rst(buckets: dict[str, list[Commit]], url: str) -> str:
    items = []
    for kind in SECTIONS:
        for c in buckets.get(kind, []):
            ref = f"`{c.sha} <{url}/commit/{c.sha}>`_" if url else f"({c.sha})"
            items.append(f"* {_line(c)} {ref}")
    return "\n".join(items)


def drop_section(existing: str, version: str) -> str:
    """``existing`` without the section of ``version`` (up to the next ``## [`` heading)."""
    v = re.escape(version.lstrip("v"))
    return re.sub(rf"^## \[{v}\].*?(?=^## \[|\Z)", "", existing, flags=re.M | re.S)


def merge_markdown(existing: str | None, section: str, version: str,
                   replace: bool = False) -> str | None:
    """The changelog with ``section`` on top, or None when ``version`` is in it (with
    ``replace`` its old section is dropped first: a release that is still being worked on)."""
    v = version.lstrip("v")
    if existing and re.search(rf"^## \[{re.escape(v)}\]", existing, flags=re.M):
        if not replace:
            return None
        existing = drop_section(existing, version)
    body = (existing or "").strip()
    if body.startswith(HEADER):
        body = body[len(HEADER):].lstrip()
    return f"{HEADER}\n\n{section}\n{body}\n".rstrip("\n") + "\n"


def merge_rst(page: str, generated: str) -> str:
    head = page.split(RST_MARKER)[0].rstrip("\n") if RST_MARKER in page else page.rstrip("\n")
    return f"{head}\n\n{RST_MARKER}\n\n{generated}\n"


def main(argv: list[str] | None = None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("version", help="version being released, e.g. 0.2.0 or v0.2.0")
    ap.add_argument("--path", default=".", help="repository root (default: current directory)")
    ap.add_argument("--since", help="tag or commit to start after (default: the latest tag)")
    ap.add_argument("--dry-run", action="store_true", help="print, write nothing")
    ap.add_argument("--replace", action="store_true",
                    help="rewrite the version's section when CHANGELOG.md already has one")
    args = ap.parse_args(argv)
    try:
        repo = Repo(args.path)
        commits = repo.commits(args.since or repo.latest_tag())
    except RuntimeError as exc:
        print(f"Error: {exc}", file=sys.stderr)
        return 1
    if not commits:
        print("No commits since the last tag: nothing to write.")
        return 0
    url = repo.web_url()
    buckets = group(commits)
    day = datetime.now(timezone.utc).strftime("%Y-%m-%d")
    section = render_markdown(args.version, buckets, url, day)
    if args.dry_run:
        print(section)
        print(render_rst(buckets, url))
        return 0
    changelog = repo.root / "CHANGELOG.md"
    merged = merge_markdown(changelog.read_text() if changelog.exists() else None, section, args.version,
                            replace=args.replace)
    if merged is None:
        print(f"CHANGELOG.md already has a section for {args.version}")
    else:
        changelog.write_text(merged)
        print(f"CHANGELOG.md: added {args.version} ({len(commits)} commits)")
    page = repo.root / "docs" / "source" / "release_notes" / f"v{args.version.lstrip('v')}.rst"
    if not page.exists():
        print(f"Warning: {page.relative_to(repo.root)} does not exist; "
              "run scripts/prepare_release.py first to create it from the template.")
        return 0
    page.write_text(merge_rst(page.read_text(), render_rst(buckets, url)))
    print(f"{page.relative_to(repo.root)}: commit list updated")
    return 0


if __name__ == "__main__":
    sys.exit(main())
