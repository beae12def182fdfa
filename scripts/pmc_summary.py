#!/usr/bin/env python
"""Summarise rocprofv3 counter-collection CSVs: per (kernel, counter) the mean value per dispatch,
the number of dispatches and the total per step (= per pass of the hot path over the batch).

    python scripts/pmc_summary.py <dir with *_counter_collection.csv, searched recursively> [steps] > pmc_summary.csv

`steps` = timed + warm-up steps of the profiled bench.py run (profile.sh: 3 + 1); kernels launched
several times per step (one launch per bucket / peak class) have per_step = mean x launches per step.
Template arguments are kept (pya_localize_kernel<true> and <false> are different kernels)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def kernel_name(raw):
    name = raw.replace("void ", "")
    m = re.match(r"([A-Za-z_0-9]+)(<[^(]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name


def main(root, steps):
    acc = defaultdict(lambda: [0, 0.0])
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                name = kernel_name(row["Kernel_Name"])
                if not name.startswith("pya_"):
                    continue
                k = (name, row["Counter_Name"])
                acc[k][0] += 1
                acc[k][1] += float(row["Counter_Value"])
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "counter", "dispatches", "mean_value", "per_step"])
    for (name, ctr), (n, s) in sorted(acc.items()):
        w.writerow([name, ctr, n, "%.1f" % (s / n), "%.1f" % (s / steps)])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else ".", float(sys.argv[2]) if len(sys.argv) > 2 else 4.0)
