set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
mkdir -p gpurun_out
echo "== headline 300 rounds"; timeout -k 10 300 python bench.py --steps 300 --warmup 3 2> gpurun_out/soak_head.err | tee gpurun_out/soak_head.json | cut -c1-400 || { tail -20 gpurun_out/soak_head.err; exit 1; }
echo "== resnet 30 rounds"; timeout -k 10 300 python -u benchmarks/run_configs.py resnet18 --rounds 30 --warmup 2 2> gpurun_out/soak_resnet.err | tee gpurun_out/soak_resnet.json | cut -c1-400 || { tail -20 gpurun_out/soak_resnet.err; exit 1; }
echo "== gpt2 10 rounds"; timeout -k 10 300 python -u benchmarks/run_configs.py gpt2_125m --rounds 10 --warmup 1 2> gpurun_out/soak_gpt2.err | tee gpurun_out/soak_gpt2.json | cut -c1-400 || { tail -20 gpurun_out/soak_gpt2.err; exit 1; }
echo "== mnist dp 30 rounds"; timeout -k 10 300 python -u benchmarks/run_configs.py mnist_cnn_dp --rounds 30 --warmup 2 2> gpurun_out/soak_dp.err | tee gpurun_out/soak_dp.json | cut -c1-400 || { tail -20 gpurun_out/soak_dp.err; exit 1; }
