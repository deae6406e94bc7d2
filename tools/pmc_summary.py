"""Sum a rocprofv3 --pmc counter over the launches of one kernel (counter_collection.csv).
usage: python tools/pmc_summary.py <dir with *counter_collection.csv> <counter> <kernel name substring>
Prints: launches, total counter value, value per launch."""
import csv
import glob
import os
import sys

root, counter, needle = sys.argv[1], sys.argv[2], sys.argv[3]
files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
tot, n = 0.0, 0
for f in files:
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] == counter and needle in row["Kernel_Name"]:
            tot += float(row["Counter_Value"])
            n += 1
print(f"{counter} {needle}: launches {n} total {tot:.6g} per_launch {tot / max(1, n):.6g}")
