# Kernel geometry (conv2 split, conv2_wgrad workgroups per client, conv1 split) against
# rounds/s at 8, 4, 2 and 1 clients on the GPU - the table behind NativeBuffers' choice.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
mkdir -p gpurun_out
run() {  # clients geometry
  NANOFED_MNIST_GEOMETRY=$2 timeout -k 10 120 python3 bench.py --gpus 1 --steps 20 --warmup 3 --clients $1 2>/dev/null \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('clients $1 geometry $2', round(d['value'],2), d['block_rounds_per_s'])" || exit 1
}
for g in 1,48,1 1,64,1 1,32,1 1,96,1 3,48,1 3,64,2 1,48,2 1,48,1; do run 8 $g; done
for g in 3,64,2 3,96,2 3,48,2 1,64,2 1,48,1 3,32,2 3,64,4 3,64,2; do run 4 $g; done
for g in 3,96,4 3,64,4 3,48,4 3,192,4 1,64,2 3,96,2 3,96,4; do run 2 $g; done
for g in 3,192,4 3,96,4 3,64,4 3,48,4 3,192,2 1,64,4 3,192,4; do run 1 $g; done
