cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
for g in 0 1 2 3; do echo "== geo $g"; NANOFED_GEMM_GEO=$g PYTHONPATH=. timeout -k 10 120 python -u tools/probes/bench_gemm_batched.py 2>&1 | grep -v "Warn\|amdgpu.ids" || exit 1; done
