set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 200 python -u tools/probes/round_breakdown.py | grep "^{" || exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 3 2> gpurun_out/hs_bench.err | cut -c1-1500 || { tail -20 gpurun_out/hs_bench.err; exit 1; }
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/hs_tests.log 2>&1 || { tail -40 gpurun_out/hs_tests.log; exit 1; }
tail -2 gpurun_out/hs_tests.log
