set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
export TMPDIR=/tmp
rm -rf gpurun_out/prof_cfg3; mkdir -p gpurun_out/prof_cfg3
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_cfg3 -o cfg3 --output-format csv -- python3 benchmarks/run_configs.py mnist_cnn_dp --rounds 4 --warmup 1 > gpurun_out/prof_cfg3/run.log 2>&1 || { tail -20 gpurun_out/prof_cfg3/run.log; exit 1; }
f=$(find gpurun_out/prof_cfg3 -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/cfg3_kernel_stats.csv
find gpurun_out/prof_cfg3 -name "*kernel_trace.csv" -delete
