#!/bin/bash
# bench.py at a given client count for several client-group sizes
cd "${GRAFT_REPO_ROOT:-.}"
clients=${1:-64}
for gsz in 2 4 8 16; do
    NANOFED_MNIST_GROUP=$gsz timeout -k 10 300 python bench.py --clients $clients --steps 3 --warmup 2 \
        --no-persist --eval-samples 2000 2>/dev/null > /tmp/gs.json || exit 1
    python - "$gsz" <<'PY'
import json, sys
d = json.load(open("/tmp/gs.json"))
print("group", sys.argv[1], round(d["value"], 2), "rounds/s", round(d["samples_per_s"] / 1e6, 2), "Msamples/s", flush=True)
PY
done
