#!/bin/bash
# Per-kernel microseconds of the native MNIST step for 1, 2 and 8 clients on the GPU, the
# headline benchmark, and (PROBE_PMC=1) a hardware-counter pass that says where the waves'
# cycles go.  Every GPU step has its own limit; a failed step ends the script.
#   tools/probes/mnist_step_probe.sh <tag>
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
tag="${1:-probe}"
out="gpurun_out/$tag"; mkdir -p "$out"
export TMPDIR=/tmp
if [ "${PROBE_TESTS:-1}" = "1" ]; then
  echo "[probe] kernel tests"; 
  timeout -k 10 600 python -u -m pytest tests/gpu/test_mnist_native_kernels.py tests/gpu/test_mnist_native_dp.py tests/gpu/test_mnist_native_batch_sizes.py \
      -x -q --timeout 120 --timeout-method thread > "$out/tests.log" 2>&1
  rc=$?; tail -3 "$out/tests.log"; [ $rc -eq 0 ] || exit $rc
fi
for g in ${PROBE_CLIENTS:-1 2 8}; do
  echo "[probe] stages, $g clients"
  timeout -k 10 200 python tools/bench_mnist_stages.py --clients $g > "$out/stages_g$g.json" 2> "$out/stages_g$g.err"
  rc=$?; cat "$out/stages_g$g.json"; [ $rc -eq 0 ] || { tail -5 "$out/stages_g$g.err"; exit $rc; }
done
if [ "${PROBE_BENCH:-1}" = "1" ]; then
  echo "[probe] bench"
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 > "$out/bench.json" 2> "$out/bench.err"
  rc=$?; cut -c1-400 "$out/bench.json"; [ $rc -eq 0 ] || { tail -5 "$out/bench.err"; exit $rc; }
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --clients 1 > "$out/bench_c1.json" 2> "$out/bench_c1.err"
  rc=$?; cut -c1-300 "$out/bench_c1.json"; [ $rc -eq 0 ] || { tail -5 "$out/bench_c1.err"; exit $rc; }
fi
if [ "${PROBE_PMC:-0}" = "1" ]; then
  cd /tmp
  root="${GRAFT_REPO_ROOT:-/root/repo}"
  rocprofv3 -L > "$root/$out/counters.txt" 2>&1 || true
  for pass in "wait:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" \
              "mem:SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
    name="${pass%%:*}"; ctrs="${pass#*:}"
    echo "[probe] pmc pass $name"
    timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv \
        -d "$root/$out/pmc_$name" -o "$name" -- python3 "$root/tools/profile_step.py" 8 4 > "$root/$out/pmc_$name.log" 2>&1
    rc=$?; [ $rc -eq 0 ] || { echo "pass $name failed $rc"; tail -5 "$root/$out/pmc_$name.log"; exit $rc; }
    python3 "$root/tools/summarize_pmc.py" "$root/$out/pmc_$name" > "$root/$out/pmc_$name.csv"
    find "$root/$out/pmc_$name" -name "*.csv" -size +2M -delete
    cat "$root/$out/pmc_$name.csv"
  done
fi
echo "[probe] done"
