set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/gpu/test_engine_gpu.py -x -q --timeout 180 --timeout-method thread > gpurun_out/prefetch_tests.log 2>&1 || { tail -40 gpurun_out/prefetch_tests.log; exit 1; }
tail -2 gpurun_out/prefetch_tests.log
for mode in "" "--no-prefetch" "" "--no-prefetch"; do
  echo "== bench $mode"
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 $mode 2> gpurun_out/prefetch_bench.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['rounds_per_s_no_persistence'],2), d['stream_set']['trial_epoch_ms'], d['final_test_accuracy'])" || { tail -20 gpurun_out/prefetch_bench.err; exit 1; }
done
