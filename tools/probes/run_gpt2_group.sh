set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/gpu/test_gpt2_native_gpu.py tests/gpu/test_gemm_kernels.py tests/gpu/test_resnet_native_gpu.py -x -q --timeout 180 --timeout-method thread 2>&1 | tail -12 || exit 1
echo "== grouped"; timeout -k 10 280 python -u benchmarks/run_configs.py gpt2_125m --rounds 3 --warmup 1 2> gpurun_out/cfg5_grouped.err | tee gpurun_out/cfg5_grouped.json || { tail -20 gpurun_out/cfg5_grouped.err; exit 1; }
echo "== client after client"; NANOFED_GPT2_GROUP=0 timeout -k 10 280 python -u benchmarks/run_configs.py gpt2_125m --rounds 3 --warmup 1 2> gpurun_out/cfg5_single.err | tee gpurun_out/cfg5_single.json || { tail -20 gpurun_out/cfg5_single.err; exit 1; }
