set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/gpu/test_mnist_native_kernels.py tests/gpu/test_mnist_native_batch_sizes.py tests/gpu/test_native_route_gpu.py tests/gpu/test_multirank_native.py -x -q --timeout 180 --timeout-method thread > gpurun_out/triple_tests.log 2>&1 || { tail -40 gpurun_out/triple_tests.log; exit 1; }
tail -2 gpurun_out/triple_tests.log
for c in 1 2 4 8; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --clients $c 2> gpurun_out/triple.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('clients $c', round(d['value'],2), round(d['rounds_per_s_no_persistence'],2), round(d['ms_per_step'],3), d['final_test_accuracy'])" || { tail -20 gpurun_out/triple.err; exit 1; }
done
