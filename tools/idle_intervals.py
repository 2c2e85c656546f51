"""Intervals during which NO kernel runs on the device, from a ``rocprofv3 --kernel-trace``
result database (``*_results.db``): where a launch-bound program leaves the GPU idle.

    python tools/idle_intervals.py out/run_results.db [--last 6000] [--min-us 15]

Prints the busy share of the analysed span, then every idle interval of at least
``--min-us`` with the kernel that ended before it and the one that started after it,
grouped by that pair (count, total, mean).
"""

from __future__ import annotations

import argparse
import sqlite3
from collections import defaultdict

from stream_gaps import short


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--last", type=int, default=0, help="only the last N dispatches")
    ap.add_argument("--min-us", type=float, default=15.0)
    args = ap.parse_args()
    cur = sqlite3.connect(args.db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "kernel_name" if "kernel_name" in cols else "display_name"
    rows = cur.execute(
        f"select d.start, d.end, s.{name_col} from rocpd_kernel_dispatch d "
        "join rocpd_info_kernel_symbol s on s.id = d.kernel_id order by d.start"
    ).fetchall()
    if args.last:
        rows = rows[-args.last:]
    span = (max(r[1] for r in rows) - rows[0][0]) / 1e3
    idle: dict[tuple[str, str], list[float]] = defaultdict(list)
    busy_until, last_name, idle_total = rows[0][1], short(rows[0][2]), 0.0
    for st, en, name in rows[1:]:
        if st > busy_until:
            gap = (st - busy_until) / 1e3
            idle_total += gap
            if gap >= args.min_us:
                idle[(last_name, short(name))].append(gap)
        if en > busy_until:
            busy_until, last_name = en, short(name)
    print(f"span_us,{span:.1f}\nidle_us,{idle_total:.1f}\nbusy_share,{1 - idle_total / span:.4f}")
    print("\nbefore,after,count,total_us,mean_us,max_us")
    for (a, b), gaps in sorted(idle.items(), key=lambda kv: -sum(kv[1])):
        print(f"{a},{b},{len(gaps)},{sum(gaps):.1f},{sum(gaps) / len(gaps):.1f},{max(gaps):.1f}")


if __name__ == "__main__":
    main()
