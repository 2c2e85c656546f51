#!/usr/bin/env bash
# Hardware-counter passes over a short program, one rocprofv3 run per pass (the
# counter blocks cannot be shared between passes).  Usage:
#   tools/pmc_passes.sh <out_dir> <python script> [args...]
# Writes <out_dir>/<pass>/... and a per-kernel summary <out_dir>/<pass>.csv
# (PMC_SUMMARY_FLAGS=--by-grid: one row per kernel and grid size).
set -uo pipefail
out="$1"; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
declare -A passes=(
  [mfma]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_WAVES"
  [lds]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT"
  [fetch]="FETCH_SIZE GRBM_GUI_ACTIVE"
  [write]="WRITE_SIZE"
)
for name in mfma lds fetch write; do
  echo "[pmc] pass $name: ${passes[$name]}"
  timeout -s KILL 150 rocprofv3 --kernel-trace --pmc ${passes[$name]} --output-format csv \
      -d "$out/$name" -o "$name" -- python3 "$@" > "$out/$name.log" 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then echo "[pmc] pass $name failed ($rc)"; tail -5 "$out/$name.log"; exit $rc; fi
  python3 "$root/tools/summarize_pmc.py" "$out/$name" ${PMC_SUMMARY_FLAGS:-} > "$out/$name.csv"
  find "$out/$name" -name "*.csv" -size +2M -delete
done
echo "[pmc] done"
