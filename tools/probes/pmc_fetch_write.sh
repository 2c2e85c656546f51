set -uo pipefail
out="$1"; prog="$2"
root="$GRAFT_REPO_ROOT"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for name in fetch write; do
  if [ $name = fetch ]; then ctr="FETCH_SIZE GRBM_GUI_ACTIVE"; else ctr="WRITE_SIZE"; fi
  timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$out/$name" -o "$name" -- python3 "$prog" > "$out/$name.log" 2>&1
  rc=$?; if [ $rc -ne 0 ]; then echo "pass $name failed ($rc)"; tail -5 "$out/$name.log"; exit $rc; fi
  python3 "$root/tools/summarize_pmc.py" "$out/$name" > "$out/$name.csv"
  find "$out/$name" -name "*.csv" -size +2M -delete
done
