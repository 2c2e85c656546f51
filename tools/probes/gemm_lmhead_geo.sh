cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
for g in default 1 2 3; do echo "== geo $g"; if [ $g = default ]; then unset NANOFED_GEMM_GEO; else export NANOFED_GEMM_GEO=$g; fi; PYTHONPATH=. timeout -k 10 200 python -u tools/probes/bench_gemm_batched.py 8 lm_head 2>&1 | grep -v "Warn\|amdgpu.ids" || exit 1; done
