#!/usr/bin/env python
"""Summarise rocprofv3 counter-collection CSVs: mean counter value per (kernel, counter).

    python scripts/pmc_summary.py <dir with *_counter_collection.csv, searched recursively> > pmc_summary.csv
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(root):
    acc = defaultdict(lambda: [0, 0.0])
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                if not name.startswith("pya_"):
                    continue
                name = name.split("<")[0]
                k = (name, row["Counter_Name"])
                acc[k][0] += 1
                acc[k][1] += float(row["Counter_Value"])
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "counter", "dispatches", "mean_value"])
    for (name, ctr), (n, s) in sorted(acc.items()):
        w.writerow([name, ctr, n, "%.1f" % (s / n)])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else ".")
