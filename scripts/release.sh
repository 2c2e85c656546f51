#!/usr/bin/env bash
# Cut a release that scripts/prepare_release.py has prepared:
#
#   scripts/release.sh            # version taken from pyproject.toml
#
# Steps: tool and branch checks, pull, changelog + release-notes commit list from
# the commits since the last tag, a pause to edit the notes, a check that the notes
# were saved after the changelog, commit, annotated tag, push.  The tag push starts
# .github/workflows/release.yml.  Set NANOFED_RELEASE_NO_PUSH=1 to stop before the push
# and NANOFED_RELEASE_YES=1 to skip the pause (CI rehearsals).
set -euo pipefail

say()  { printf '==> %s\n' "$1"; }
fail() { printf 'Error: %s\n' "$1" >&2; exit 1; }
need() { command -v "$1" >/dev/null 2>&1 || fail "required command '$1' not found"; }

need git
need python
root="$(git rev-parse --show-toplevel)"
cd "$root"

branch="$(git branch --show-current)"
[ "$branch" = "main" ] || fail "releases are cut from main (currently on '$branch')"
if git remote get-url origin >/dev/null 2>&1 && [ "${NANOFED_RELEASE_NO_PUSH:-0}" != "1" ]; then
    say "Pulling main"
    git pull --ff-only origin main
fi

version="$(python - <<'PY'
import re
print(re.search(r'(?m)^version\s*=\s*"([^"]+)"', open("pyproject.toml").read()).group(1))
PY
)"
pkg_version="$(python -c 'import re;print(re.search(r"__version__\s*=\s*\"([^\"]+)\"", open("nanofed_amd/__init__.py").read()).group(1))')"
[ "$version" = "$pkg_version" ] || fail "pyproject.toml says $version, nanofed_amd/__init__.py says $pkg_version"
if git rev-parse -q --verify "refs/tags/v$version" >/dev/null; then
    fail "tag v$version exists already: run scripts/prepare_release.py <new version> first"
fi
notes="docs/source/release_notes/v$version.rst"
[ -f "$notes" ] || fail "$notes not found: run scripts/prepare_release.py $version first"

say "Generating the changelog for v$version"
python scripts/changelog.py "$version"

if [ "${NANOFED_RELEASE_YES:-0}" != "1" ]; then
    printf '\nReview and complete the release notes now:\n  %s\nPress Enter when they are saved... ' "$notes"
    read -r _
    mtime() { stat -c %Y "$1" 2>/dev/null || stat -f %m "$1"; }
    [ "$(mtime "$notes")" -ge "$(mtime CHANGELOG.md)" ] || fail "the release notes were not saved after the changelog was generated"
fi

say "Committing"
git add -A
git commit -m "chore: prepare release v$version"
say "Tagging v$version"
git tag -a "v$version" -m "Release v$version"
if [ "${NANOFED_RELEASE_NO_PUSH:-0}" = "1" ]; then
    say "NANOFED_RELEASE_NO_PUSH=1: tag created locally, nothing pushed"
    exit 0
fi
git push origin main
git push origin "v$version"
say "Released v$version"
printf 'Next: watch the Release workflow, check the published wheel, check the documentation build.\n'
