"""Per-kernel means of rocprofv3 --pmc counter CSV output.

    python tools/summarize_pmc.py <dir> [--by-grid]

``--by-grid`` keeps launches of one kernel with different grid sizes apart
(one row per shape).  Kernel names that rocprofv3 leaves mangled (bf16 pointer
arguments) are reduced to ``name<template ints>``.
"""
import csv
import glob
import re
import sys
from collections import defaultdict


def short_name(raw: str) -> str:
    m = re.match(r"_ZN\d+_GLOBAL__N_1(\d+)", raw)
    if m:
        n, start = int(m.group(1)), m.end()
        ident, rest = raw[start : start + n], raw[start + n :]
        tm = re.match(r"I((?:L[ib]\d+E)+)E", rest)
        if tm:
            ident += "<" + ",".join(re.findall(r"L[ib](\d+)E", tm.group(1))) + ">"
        return ident
    return raw.replace("(anonymous namespace)::", "").split("(")[0]


def main() -> None:
    by_grid = "--by-grid" in sys.argv
    root = [a for a in sys.argv[1:] if not a.startswith("--")][0]
    path = glob.glob(root + "/**/*counter_collection.csv", recursive=True)[0]
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(path)):
        name = short_name(row["Kernel_Name"])
        if by_grid:
            name += f"@{row.get('Grid_Size', '?')}"
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    counters = sorted({c for k in acc.values() for c in k})
    print("kernel," + ",".join(counters))
    for k, d in sorted(acc.items()):
        base = k.split("@")[0]
        if not base.endswith("_kernel") and "_kernel<" not in base:
            continue
        label = k if len(k) <= 48 else k[-48:]
        label = '"' + label + '"' if "," in label else label
        print(label + "," + ",".join(f"{sum(d[c]) / max(len(d[c]), 1):.0f}" for c in counters))


if __name__ == "__main__":
    main()
