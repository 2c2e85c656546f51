"""Per-step kernel breakdown of the steady state from a rocprofv3 kernel trace.

    python tools/steady_state_kernels.py <kernel_trace.csv> [--marker sgd_step_kernel] [--steps 256]

The window is the last ``--steps`` occurrences of the marker kernel (one per
optimisation step), so library autotuning and warm-up are left out.
"""

from __future__ import annotations

import argparse
import csv
import json
from collections import defaultdict


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--marker", default="sgd_step_kernel")
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--top", type=int, default=45)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    rows = []
    with open(args.trace, newline="") as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    marks = sorted(s for s, _, n in rows if args.marker in n)
    steps = min(args.steps, len(marks) - 1)
    lo, hi = marks[-steps - 1], marks[-1]
    acc: dict[str, list[int]] = defaultdict(lambda: [0, 0])
    for s, e, n in rows:
        if lo < s <= hi:
            acc[n[:110]][0] += 1
            acc[n[:110]][1] += e - s
    total = sum(v[1] for v in acc.values())
    print(f"steps {steps}  window {(hi - lo) / 1e6:.1f} ms  busy {total / 1e6:.1f} ms  "
          f"busy per step {total / steps / 1e3:.1f} us")
    table = []
    for name, (calls, ns) in sorted(acc.items(), key=lambda kv: -kv[1][1])[: args.top]:
        table.append({"kernel": name, "calls_per_step": round(calls / steps, 2),
                      "us_per_step": round(ns / steps / 1e3, 1),
                      "us_per_call": round(ns / calls / 1e3, 1),
                      "percent": round(100.0 * ns / total, 1)})
        print(f"{name:110s} {calls / steps:6.1f}/step {ns / steps / 1e3:8.1f} us/step "
              f"{ns / calls / 1e3:7.1f} us {100.0 * ns / total:5.1f}%")
    if args.json:
        with open(args.json, "w") as fh:
            json.dump({"steps": steps, "busy_us_per_step": total / steps / 1e3, "kernels": table}, fh, indent=1)


if __name__ == "__main__":
    main()
