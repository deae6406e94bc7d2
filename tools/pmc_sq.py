"""Summarise SQ counters of one kernel from a rocprofv3 --pmc run (counter_collection.csv).
usage: python tools/pmc_sq.py <dir> <kernel substring>"""
import csv
import glob
import os
import sys
from collections import defaultdict

root, needle = sys.argv[1], sys.argv[2]
tot = defaultdict(float)
files = sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
for f in files[-1:]:  # the newest run only (gpurun merges a re-run's files into the same directory: summing them all doubled the counts)
    for row in csv.DictReader(open(f)):
        if needle in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
for k in sorted(tot):
    print(f"{k:28s} {tot[k]:.6g}")
if "SQ_WAVE_CYCLES" in tot:
    w = tot["SQ_WAVE_CYCLES"]
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM",
              "SQ_ACTIVE_INST_SCA", "SQ_WAIT_INST_LDS"):
        if k in tot:
            print(f"  {k} / SQ_WAVE_CYCLES = {tot[k] / w:.3f}")
