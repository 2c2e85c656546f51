set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_cfg4g
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_cfg4g -o cfg4g --output-format csv -- python3 benchmarks/run_configs.py resnet18 --rounds 6 --warmup 2 > gpurun_out/prof_cfg4g/run.log 2>&1 || { tail -20 gpurun_out/prof_cfg4g/run.log; exit 1; }
tail -2 gpurun_out/prof_cfg4g/run.log
find gpurun_out/prof_cfg4g -name "*kernel_stats.csv" | head
f=$(find gpurun_out/prof_cfg4g -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/cfg4_grouped_kernel_stats.csv
find gpurun_out/prof_cfg4g -name "*kernel_trace.csv" -delete
