# A/B: switch named in the variable: on (1) or off (0).
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
for v in 1 0 1 0 1 0; do
  echo -n "NANOFED_RESNET_JOIN=$v  "
  NANOFED_RESNET_JOIN=$v timeout -k 10 200 python3 benchmarks/run_configs.py resnet18 --rounds 10 --warmup 3 \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['rounds_per_s'], d['train_loss'])" || exit 1
done
