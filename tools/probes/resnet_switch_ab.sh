# A/B of one switch of the grouped ResNet step (config 4, 8 clients) on one box:
#   resnet_switch_ab.sh NANOFED_CONV_XORDER      (on = 1 / unset, off = 0)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
var=${1:-NANOFED_RESNET_JOIN}
for v in on off on off on off; do
  echo -n "$var $v  "
  if [ $v = off ]; then export $var=0; else unset $var; fi
  timeout -k 10 200 python3 benchmarks/run_configs.py resnet18 --rounds 10 --warmup 3 \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['rounds_per_s'], d['train_loss'])" || exit 1
done
