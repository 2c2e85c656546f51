# A/B of the BatchNorm backward folded into the data gradients (config 4, 8 clients).
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
mkdir -p gpurun_out
for v in 1 0 1 0; do
  echo -n "NANOFED_RESNET_FUSE_BN_BWD=$v  "
  NANOFED_RESNET_FUSE_BN_BWD=$v timeout -k 10 200 python3 benchmarks/run_configs.py resnet18 --rounds 10 --warmup 3 \
    | tee -a gpurun_out/fused_bn_ab.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['rounds_per_s'], d['train_loss'], d['eval']['accuracy'])" || exit 1
done
