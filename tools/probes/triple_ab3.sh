set -o pipefail
cd "${GRAFT_REPO_ROOT}"
timeout -k 10 400 python -u -m pytest tests/gpu/test_mnist_native_kernels.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -2 || exit 1
run() {
  NANOFED_MNIST_TRIPLE=$2 timeout -k 10 120 python3 bench.py --gpus 1 --steps 20 --warmup 3 --clients $1 2>/dev/null \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('clients $1 triple $2', round(d['value'],2), d['block_rounds_per_s'], d['final_test_accuracy'])" || exit 1
}
run 1 1; run 1 0; run 1 1; run 2 1; run 2 0; run 4 1; run 4 0; run 8 1; run 8 0
timeout -k 10 120 python3 tools/bench_mnist_stages.py --clients 1 2>/dev/null | tail -1
