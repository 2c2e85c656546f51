set -o pipefail
cd "${GRAFT_REPO_ROOT}"
run() {
  NANOFED_MNIST_GEOMETRY=$2 timeout -k 10 120 python3 bench.py --gpus 1 --steps 20 --warmup 3 --clients $1 2>/dev/null \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('clients $1 geometry $2', round(d['value'],2), d['block_rounds_per_s'], d['final_test_accuracy'])" || exit 1
}
for g in 1,48,1 1,24,1 1,32,2 1,48,4 1,16,1 1,64,1 1,48,1; do run 8 $g; done
for g in 3,192,2 3,192,1 3,96,2 3,192,4 3,192,2; do run 1 $g; done
