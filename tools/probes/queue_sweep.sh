#!/bin/bash
# bench.py under different hardware-queue counts, group sizes and stream counts
cd "${GRAFT_REPO_ROOT:-.}"
for spec in "4 2 4" "8 2 4" "8 1 8" "8 2 8" "16 1 8" "6 2 4"; do
    set -- $spec
    echo "queues=$1 group=$2 streams=$3"
    GPU_MAX_HW_QUEUES=$1 NANOFED_MNIST_GROUP=$2 NANOFED_MNIST_STREAMS=$3 \
        timeout -k 10 120 python bench.py --steps 15 --warmup 2 --no-persist 2>/dev/null \
        | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(round(d['value'], 2), d['stream_set'])" || exit 1
done
