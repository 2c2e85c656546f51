set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== conv + resnet tests"; 
timeout -k 10 600 python -u -m pytest tests/gpu/test_conv_kernels.py tests/gpu/test_resnet_native_gpu.py -x -q --timeout 120 --timeout-method thread > gpurun_out/conv_pad_tests.log 2>&1 || { tail -30 gpurun_out/conv_pad_tests.log; exit 1; }
tail -3 gpurun_out/conv_pad_tests.log
echo "== bench_conv batch 512"
timeout -k 10 300 python -u tools/bench_conv.py --batch 512 --no-library > gpurun_out/conv_pad_bench512.log 2>&1 || { tail -30 gpurun_out/conv_pad_bench512.log; exit 1; }
tail -12 gpurun_out/conv_pad_bench512.log
echo "== config 4"
timeout -k 10 400 python -u benchmarks/run_configs.py resnet18 --rounds 6 --warmup 2 > gpurun_out/conv_pad_cfg4.log 2>&1 || { tail -30 gpurun_out/conv_pad_cfg4.log; exit 1; }
tail -2 gpurun_out/conv_pad_cfg4.log
