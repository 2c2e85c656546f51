"""Per-kernel means of rocprofv3 --pmc counter CSV output."""
import csv
import glob
import sys
from collections import defaultdict

path = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = defaultdict(lambda: defaultdict(list))
for row in csv.DictReader(open(path)):
    name = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
counters = sorted({c for k in acc.values() for c in k})
print("kernel," + ",".join(counters))
for k, d in sorted(acc.items()):
    if not k.endswith("_kernel") and "_kernel<" not in k:
        continue
    print(k[-40:] + "," + ",".join(f"{sum(d[c]) / max(len(d[c]), 1):.0f}" for c in counters))
