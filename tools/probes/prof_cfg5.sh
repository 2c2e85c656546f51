set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_cfg5g
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_cfg5g -o cfg5g --output-format csv -- python3 benchmarks/run_configs.py gpt2_125m --rounds 3 --warmup 1 > gpurun_out/prof_cfg5g/run.log 2>&1 || { tail -20 gpurun_out/prof_cfg5g/run.log; exit 1; }
f=$(find gpurun_out/prof_cfg5g -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/cfg5_grouped_kernel_stats.csv
find gpurun_out/prof_cfg5g -name "*kernel_trace.csv" -delete
