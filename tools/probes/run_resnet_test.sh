set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests/gpu/test_resnet_native_gpu.py -x -v --timeout 120 --timeout-method thread > gpurun_out/resnet_native_test.log 2>&1
rc=$?
tail -40 gpurun_out/resnet_native_test.log
exit $rc
