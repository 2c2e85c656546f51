"""Per-kernel statistics from a ``rocprofv3 --kernel-trace`` result database
(``*_results.db``, the rocpd SQLite format): calls, total / mean / min / max
duration and share of the kernel time, as CSV on stdout.

    python tools/rocpd_kernel_stats.py gpurun_out/prof/x_results.db [--top 40] > profiles/x_kernel_stats.csv
"""

from __future__ import annotations

import argparse
import csv
import sqlite3
import sys


def kernel_stats(path: str) -> list[dict]:
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
    sym_cols = [r[1] for r in cur.execute("pragma table_info(rocpd_info_kernel_symbol)")]
    name_col = "kernel_name" if "kernel_name" in sym_cols else "display_name"
    if not {"start", "end", "kernel_id"} <= set(cols):
        raise SystemExit(f"unexpected rocpd_kernel_dispatch columns: {cols}")
    rows = cur.execute(
        f"select s.{name_col}, count(*), sum(d.end - d.start), min(d.end - d.start), "
        "max(d.end - d.start) from rocpd_kernel_dispatch d "
        "join rocpd_info_kernel_symbol s on s.id = d.kernel_id group by 1 order by 3 desc"
    ).fetchall()
    total = sum(r[2] for r in rows) or 1
    return [
        {"Name": r[0], "Calls": r[1], "TotalDurationNs": r[2], "AverageNs": r[2] / r[1],
         "Percentage": 100.0 * r[2] / total, "MinNs": r[3], "MaxNs": r[4]}
        for r in rows
    ]


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--top", type=int, default=0, help="keep only the first N rows")
    ap.add_argument("--max-name", type=int, default=160, help="cut kernel names to this length")
    args = ap.parse_args()
    rows = kernel_stats(args.db)
    if args.top:
        rows = rows[: args.top]
    w = csv.DictWriter(sys.stdout, fieldnames=list(rows[0].keys()) if rows else ["Name"])
    w.writeheader()
    for r in rows:
        r["Name"] = r["Name"][: args.max_name]
        r["AverageNs"] = round(r["AverageNs"], 1)
        r["Percentage"] = round(r["Percentage"], 3)
        w.writerow(r)


if __name__ == "__main__":
    main()
