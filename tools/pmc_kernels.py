#!/usr/bin/env python3
"""Per-kernel sums of the counters of one or more rocprofv3 --pmc runs: python tools/pmc_kernels.py <dir> [<dir> ...]
(one line per kernel: dispatches and every counter found; kernel names shortened)"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

tot = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for root in sys.argv[1:]:
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void azd::", "").replace("azd::", "")
            tot[name][row["Counter_Name"]] += float(row["Counter_Value"])
            disp[name].add((f, row.get("Dispatch_Id")))
for name in sorted(tot, key=lambda n: -sum(tot[n].values())):
    print("%-60s dispatches %6d  %s" % (name[:60], len(disp[name]), "  ".join("%s %.4g" % (k, v) for k, v in sorted(tot[name].items()))))
