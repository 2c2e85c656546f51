# Where the waves of the GEMM kernels spend their cycles at 4096^3: two counter passes
# (wait / issue classes), one rocprofv3 run each.  Usage: gemm_pmc_wait.sh <abs out dir>
set -uo pipefail
out="$1"
root="$(cd "$(dirname "$0")/../.." && pwd)"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
declare -A passes=(
  [wait]="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
  [mem]="SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES"
)
for name in wait mem; do
  timeout -s KILL 150 rocprofv3 --kernel-trace --pmc ${passes[$name]} --output-format csv \
      -d "$out/$name" -o "$name" -- python3 "$root/tools/probes/gemm_pmc_probe.py" 4096 > "$out/$name.log" 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then echo "pass $name failed ($rc)"; tail -5 "$out/$name.log"; exit $rc; fi
  python3 "$root/tools/summarize_pmc.py" "$out/$name" > "$out/$name.csv"
  find "$out/$name" -name "*.csv" -size +2M -delete
done
