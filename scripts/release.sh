#!/usr/bin/env bash
# Tag and push a prepared release:  scripts/release.sh X.Y.Z
set -euo pipefail
version="${1:?usage: release.sh X.Y.Z}"
branch="$(git rev-parse --abbrev-ref HEAD)"
if [ "$branch" != "main" ]; then echo "releases are cut from main (on $branch)" >&2; exit 1; fi
grep -q "version = \"$version\"" pyproject.toml || { echo "run scripts/prepare_release.py $version first" >&2; exit 1; }
git add -A
git commit -m "chore: release $version" || true
git tag -a "v$version" -m "Release $version"
git push origin main "v$version"
echo "released v$version"
