set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/gpu/test_bn_kernels.py tests/gpu/test_resnet_native_gpu.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -4 || exit 1
echo "== native grouped"; timeout -k 10 280 python -u benchmarks/run_configs.py resnet18 --rounds 6 --warmup 2 2> gpurun_out/cfg4_grouped.err | tee gpurun_out/cfg4_grouped.json || { tail -20 gpurun_out/cfg4_grouped.err; exit 1; }
echo "== per-client trainer"; NANOFED_RESNET_NATIVE=0 timeout -k 10 280 python -u benchmarks/run_configs.py resnet18 --rounds 6 --warmup 2 2> gpurun_out/cfg4_perclient.err | tee gpurun_out/cfg4_perclient.json || { tail -20 gpurun_out/cfg4_perclient.err; exit 1; }
