# Per-kernel times of the 3x3 convolution passes at batch 512 (rocprofv3 kernel trace).
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_conv
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_conv -o conv --output-format csv -- python3 tools/bench_conv.py --batch 512 --no-library --reps 20 > gpurun_out/prof_conv/run.log 2>&1 || { tail -20 gpurun_out/prof_conv/run.log; exit 1; }
find gpurun_out/prof_conv -name "*kernel_trace.csv" -delete
cp gpurun_out/prof_conv/conv_kernel_stats.csv gpurun_out/conv_kernel_stats.csv
