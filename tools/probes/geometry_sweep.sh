#!/bin/bash
# bench.py (8 clients, no persistence) over client-group sizes and kernel geometries:
#   tools/probes/geometry_sweep.sh "<group>:<split,wpc,conv1_split> ..."
# group = clients per launch chain; the geometry triple is NANOFED_MNIST_GEOMETRY.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for cfg in ${1:-"2:1,64,1"}; do
    grp="${cfg%%:*}"; geo="${cfg#*:}"
    NANOFED_MNIST_GROUP=$grp NANOFED_MNIST_GEOMETRY=$geo timeout -k 10 200 python bench.py \
        --steps ${SWEEP_STEPS:-10} --warmup 3 --no-persist 2>/dev/null > /tmp/gs.json || { echo "group $grp geometry $geo failed"; exit 1; }
    python - "$grp" "$geo" <<'PY'
import json, sys
d = json.load(open("/tmp/gs.json"))
print("group", sys.argv[1], "geometry", sys.argv[2], round(d["value"], 2), "rounds/s",
      d.get("stream_set", {}).get("trial_epoch_ms"), flush=True)
PY
done
