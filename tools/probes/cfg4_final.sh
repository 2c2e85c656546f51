# Config 4 (ResNet-18, 8 clients) at both shard sizes, then a kernel table of the 1,024-sample run.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
mkdir -p gpurun_out
timeout -k 10 200 python3 benchmarks/run_configs.py resnet18 --rounds 10 --warmup 3 > gpurun_out/cfg4_1024.json || exit 1
tail -c 600 gpurun_out/cfg4_1024.json; echo
timeout -k 10 400 python3 benchmarks/run_configs.py resnet18 --rounds 5 --warmup 2 --samples-per-client 6250 > gpurun_out/cfg4_6250.json || exit 1
tail -c 600 gpurun_out/cfg4_6250.json; echo
bash tools/probes/prof_cfg4.sh
