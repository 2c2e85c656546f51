"""Idle time between consecutive kernels of the same hardware queue, from a
``rocprofv3 --kernel-trace`` result database (``*_results.db``).

    python tools/stream_gaps.py out/run_results.db [--kernel conv2_fwd] [--last 4000]

For a launch chain (every kernel depends on the one before) the gap between one
kernel's end and the next one's start on the same queue is pure dependency / launch
latency.  The table lists, per (previous kernel → next kernel) pair, the number of
occurrences and the mean / median gap, then per queue the busy share
(kernel time / span).  ``--last N`` restricts the analysis to the last N dispatches
(the steady state of a benchmark run).
"""

from __future__ import annotations

import argparse
import re
import sqlite3
import statistics
from collections import defaultdict


def short(name: str) -> str:
    m = re.search(r"\d+([A-Za-z_0-9]+?_kernel)", name)  # mangled: ..._N_116conv2_fwd_kernelE...
    if m:
        return m.group(1)
    name = name.split("(")[0]
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("::")[-1][:28]


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--last", type=int, default=0)
    ap.add_argument("--max-gap-us", type=float, default=200.0,
                    help="gaps longer than this are breaks between chains, not dependency latency")
    args = ap.parse_args()
    cur = sqlite3.connect(args.db).cursor()
    sym_cols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "kernel_name" if "kernel_name" in sym_cols else "display_name"
    rows = cur.execute(
        f"select d.queue_id, d.start, d.end, s.{name_col} from rocpd_kernel_dispatch d "
        "join rocpd_info_kernel_symbol s on s.id = d.kernel_id order by d.start"
    ).fetchall()
    if args.last:
        rows = rows[-args.last:]
    per_queue: dict[int, list] = defaultdict(list)
    for q, st, en, name in rows:
        per_queue[q].append((st, en, short(name)))
    pairs: dict[tuple[str, str], list[float]] = defaultdict(list)
    print("queue,kernels,span_ms,busy_ms,busy_share")
    for q, ks in sorted(per_queue.items()):
        busy = sum(en - st for st, en, _ in ks) / 1e6
        span = (ks[-1][1] - ks[0][0]) / 1e6
        print(f"{q},{len(ks)},{span:.3f},{busy:.3f},{busy / span if span else 0:.3f}")
        for (s0, e0, n0), (s1, e1, n1) in zip(ks, ks[1:]):
            gap = (s1 - e0) / 1e3
            if gap <= args.max_gap_us:
                pairs[(n0, n1)].append(gap)
    print("\nprevious,next,count,mean_gap_us,median_gap_us,p90_gap_us")
    for (a, b), gaps in sorted(pairs.items(), key=lambda kv: -len(kv[1])):
        if len(gaps) < 20:
            continue
        gaps.sort()
        print(f"{a},{b},{len(gaps)},{statistics.fmean(gaps):.2f},{statistics.median(gaps):.2f},"
              f"{gaps[int(0.9 * (len(gaps) - 1))]:.2f}")


if __name__ == "__main__":
    main()
