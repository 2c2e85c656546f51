#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
for spec in "4 1" "4 2" "4 4" "6 2" "6 3" "2 1" "2 2"; do
    set -- $spec
    NANOFED_MNIST_GROUP=$2 timeout -k 10 300 python bench.py --clients $1 --steps 8 --warmup 2 \
        --no-persist --eval-samples 2000 2>/dev/null > /tmp/gs.json || exit 1
    python - "$1" "$2" <<'PY'
import json, sys
d = json.load(open("/tmp/gs.json"))
print("clients", sys.argv[1], "group", sys.argv[2], round(d["value"], 2), "rounds/s", d["stream_set"], flush=True)
PY
done
