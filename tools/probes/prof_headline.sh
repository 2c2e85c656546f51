set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
export TMPDIR=/tmp
rm -rf gpurun_out/prof_head; mkdir -p gpurun_out/prof_head
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_head -o head --output-format csv -- python3 bench.py --steps 6 --warmup 3 --no-persist > gpurun_out/prof_head/run.log 2>&1 || { tail -20 gpurun_out/prof_head/run.log; exit 1; }
tail -1 gpurun_out/prof_head/run.log | cut -c1-200
f=$(find gpurun_out/prof_head -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/headline_kernel_stats.csv
find gpurun_out/prof_head -name "*kernel_trace.csv" -delete
head -16 gpurun_out/headline_kernel_stats.csv | cut -c1-160
