set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 python -u tools/probes/stall_probe2.py 4 | grep "^total\|^gc\|_host_indices" | cut -c1-330 || exit 1
for c in 8 4 2 1; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --clients $c 2> gpurun_out/bc.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('clients $c', round(d['value'],2), round(d['rounds_per_s_no_persistence'],2), round(d['ms_per_step'],3), d['final_test_accuracy'])" || { tail -20 gpurun_out/bc.err; exit 1; }
done
timeout -k 10 600 python -u -m pytest tests/gpu/test_mnist_native_kernels.py tests/gpu/test_mnist_native_dp.py tests/gpu/test_mnist_native_batch_sizes.py tests/gpu/test_engine_gpu.py tests/gpu/test_multirank_native.py tests/gpu/test_stream_ordering.py -x -q --timeout 180 --timeout-method thread > gpurun_out/hs_tests.log 2>&1 || { tail -40 gpurun_out/hs_tests.log; exit 1; }
tail -2 gpurun_out/hs_tests.log
