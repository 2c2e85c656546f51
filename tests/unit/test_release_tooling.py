"""Release tooling: changelog generation in a scratch git repository, version
rules and release-notes creation of the release gate."""

import importlib.util
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[2]


def _load(name):
    spec = importlib.util.spec_from_file_location(name, ROOT / "scripts" / f"{name}.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


changelog = _load("changelog")
prepare = _load("prepare_release")


def _git(repo, *args):
    subprocess.run(["git", *args], cwd=repo, check=True, capture_output=True,
                   env={"GIT_AUTHOR_NAME": "t", "GIT_AUTHOR_EMAIL": "t@t", "GIT_COMMITTER_NAME": "t",
                        "GIT_COMMITTER_EMAIL": "t@t", "PATH": "/usr/bin:/bin", "HOME": str(repo)})


@pytest.fixture()
def repo(tmp_path):
    _git(tmp_path, "init", "-q", "-b", "main")
    for i, subject in enumerate([
        "feat(engine): device coordinator",
        "fix: join the writer thread on close",
        "perf(gemm)!: new tile order",
        "tests: cover the watchdog",
        "Plain subject without a type",
    ]):
        (tmp_path / f"f{i}").write_text(str(i))
        _git(tmp_path, "add", "-A")
        _git(tmp_path, "commit", "-q", "-m", subject)
    return tmp_path


def test_subjects_are_parsed_and_grouped():
    c = changelog.parse("abc1234", "perf(gemm)!: new tile order")
    assert (c.kind, c.scope, c.text, c.breaking) == ("perf", "gemm", "new tile order", True)
    assert changelog.parse("a", "tests: x").kind == "test"
    assert changelog.parse("a", "Merge stuff: like this").kind == "other"
    assert changelog.parse("a", "no type here").text == "no type here"


def test_changelog_and_release_notes_are_written_once(repo, capsys):
    notes = repo / "docs" / "source" / "release_notes"
    notes.mkdir(parents=True)
    (notes / "v0.2.0.rst").write_text("Version 0.2.0\n=============\n\nOverview\n--------\nhand-written\n\n"
                                      f"Changelog\n---------\n{changelog.RST_MARKER}\n\n* stale entry\n")
    _git(repo, "remote", "add", "origin", "git@github.com:owner/name.git")
    assert changelog.main(["0.2.0", "--path", str(repo)]) == 0
    md = (repo / "CHANGELOG.md").read_text()
    assert md.startswith("# Changelog\n\n## [0.2.0](https://github.com/owner/name/releases/tag/v0.2.0) - ")
    assert md.index("### Features") < md.index("### Bug Fixes") < md.index("### Performance") \
        < md.index("### Tests") < md.index("### Other Changes")
    assert "- (engine) device coordinator ([" in md and "(gemm) new tile order [BREAKING]" in md
    assert "https://github.com/owner/name/commit/" in md
    rst = (notes / "v0.2.0.rst").read_text()
    assert "hand-written" in rst and "stale entry" not in rst
    assert rst.count(changelog.RST_MARKER) == 1 and "* (engine) device coordinator `" in rst
    # a second run leaves the changelog alone and rewrites the same list
    assert changelog.main(["0.2.0", "--path", str(repo)]) == 0
    assert (repo / "CHANGELOG.md").read_text() == md
    assert (notes / "v0.2.0.rst").read_text() == rst
    # an older section stays below a newer one
    assert changelog.main(["v0.3.0", "--path", str(repo)]) == 0
    md2 = (repo / "CHANGELOG.md").read_text()
    assert md2.index("## [0.3.0]") < md2.index("## [0.2.0]") and md2.count("# Changelog") == 1
    assert "does not exist" in capsys.readouterr().out  # no v0.3.0.rst page yet


def test_replace_rewrites_the_section_of_a_release_in_progress():
    old = (f"{changelog.HEADER}\n\n## [0.2.0] - 2026-01-01\n\n- first list\n\n"
           "## [0.1.0] - 2025-01-01\n\n- older\n")
    new_section = "## [0.2.0] - 2026-02-02\n\n- second list\n"
    assert changelog.merge_markdown(old, new_section, "0.2.0") is None
    out = changelog.merge_markdown(old, new_section, "v0.2.0", replace=True)
    assert out.count("## [0.2.0]") == 1 and "second list" in out and "first list" not in out
    assert "## [0.1.0] - 2025-01-01" in out and "- older" in out
    assert out.index("## [0.2.0]") < out.index("## [0.1.0]")
    # the last section of the file can be the one that is replaced
    only = f"{changelog.HEADER}\n\n## [0.2.0] - 2026-01-01\n\n- first list\n"
    out = changelog.merge_markdown(only, new_section, "0.2.0", replace=True)
    assert "first list" not in out and "second list" in out


def test_only_commits_after_the_latest_tag_are_listed(repo):
    _git(repo, "tag", "v0.1.0")
    (repo / "new").write_text("x")
    _git(repo, "add", "-A")
    _git(repo, "commit", "-q", "-m", "docs: explain the tail graph")
    r = changelog.Repo(repo)
    assert r.latest_tag() == "v0.1.0"
    commits = r.commits(r.latest_tag())
    assert [c.text for c in commits] == ["explain the tail graph"]
    assert r.web_url() == ""


def test_version_rules():
    key = prepare.version_key
    assert key("0.2.0") > key("0.2.0-rc.2") > key("0.2.0-rc.1") > key("0.2.0-beta.9") > key("0.1.9")
    for bad in ("1.0", "v1.0.0", "1.0.0-dev", "1.0.0.0"):
        with pytest.raises(ValueError):
            key(bad)
    assert prepare.SEMVER.match(prepare.current_version())


def test_release_notes_come_from_the_template(tmp_path, monkeypatch):
    notes = tmp_path / "docs" / "source" / "release_notes"
    notes.mkdir(parents=True)
    (notes / "template.rst").write_text((ROOT / "docs/source/release_notes/template.rst").read_text())
    (notes / "index.rst").write_text("Release notes\n=============\n\n.. toctree::\n   :maxdepth: 1\n\n   v0.1.0\n")
    monkeypatch.setattr(prepare, "NOTES_DIR", notes)
    monkeypatch.setattr(prepare, "ROOT", tmp_path)
    page = prepare.create_release_notes("0.2.0")
    text = page.read_text()
    assert text.startswith("Version 0.2.0\n") and "{version}" not in text and "{date}" not in text
    assert changelog.RST_MARKER in text
    index = (notes / "index.rst").read_text()
    assert index.index("v0.2.0") < index.index("v0.1.0")
    page.write_text("edited")
    assert prepare.create_release_notes("0.2.0").read_text() == "edited"  # never overwritten


def test_the_gate_refuses_bad_arguments(capsys):
    assert prepare.main(["not-a-version"]) == 2
    assert prepare.main([prepare.current_version()]) == 2  # not greater than the current one
    assert prepare.main(["99.0.0", "--skip", "nonsense"]) == 2
