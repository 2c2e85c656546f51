set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for args in "--steps 20 --warmup 3" "--steps 20 --warmup 3 --no-persist" "--steps 20 --warmup 3" "--steps 40 --warmup 3" "--steps 10 --warmup 2"; do
  timeout -k 10 300 python bench.py $args 2> gpurun_out/br.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$args', round(d['value'],2), d['rounds_per_s_no_persistence'] and round(d['rounds_per_s_no_persistence'],2), round(d['ms_per_step'],3), d['final_test_accuracy'])" || { tail -20 gpurun_out/br.err; exit 1; }
done
