set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in 8 4 2 1; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --clients $c 2> gpurun_out/bc.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('clients $c', round(d['value'],2), round(d['rounds_per_s_no_persistence'],2), round(d['ms_per_step'],3), d['final_test_accuracy'])" || { tail -20 gpurun_out/bc.err; exit 1; }
done
