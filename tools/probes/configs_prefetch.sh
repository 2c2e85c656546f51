set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in mnist_cnn_dp resnet18 gpt2_125m; do
  echo "== $c"
  timeout -k 10 500 python -u benchmarks/run_configs.py $c --rounds 6 --warmup 2 > gpurun_out/cfgp_$c.log 2>&1 || { tail -30 gpurun_out/cfgp_$c.log; exit 1; }
  tail -1 gpurun_out/cfgp_$c.log | cut -c1-330
done
