#!/bin/bash
# Full GPU check of a tree: the GPU test suite, the headline benchmark, and a kernel
# trace of the reference-style in-node example.  Every GPU step has its own limit and
# a failed step ends the script.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread \
    > gpurun_out/gpu_tests.log 2>&1
rc=$?; tail -3 gpurun_out/gpu_tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err
rc=$?; cat gpurun_out/bench.json; [ $rc -eq 0 ] || { tail -5 gpurun_out/bench.err; exit $rc; }
rm -rf gpurun_out/prof_innode
timeout -k 10 400 rocprofv3 --kernel-trace -d gpurun_out/prof_innode -o innode -- \
    python3 examples/mnist/run_experiment.py innode > gpurun_out/innode.log 2>&1
rc=$?; tail -2 gpurun_out/innode.log; [ $rc -eq 0 ] || exit $rc
db=$(find gpurun_out/prof_innode -name '*_results.db' | head -1)
python tools/rocpd_kernel_stats.py "$db" --top 40 > gpurun_out/innode_kernel_stats.csv
head -12 gpurun_out/innode_kernel_stats.csv
