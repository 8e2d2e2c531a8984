#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per kernel name.
    python tools/pmc_summary.py <dir-or-csv> [<dir-or-csv> ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def collect(path):
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection*.csv"), recursive=True)
    acc = defaultdict(list)
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                acc[(row["Kernel_Name"], row["Counter_Name"])].append(float(row["Counter_Value"]))
    return acc


def main(paths):
    for p in paths:
        for (kern, ctr), vals in sorted(collect(p).items()):
            k = kern if len(kern) < 110 else kern[:107] + "..."
            print("%-14s mean %14.1f  min %14.1f  max %14.1f  n=%4d  %s" % (ctr, sum(vals) / len(vals), min(vals), max(vals), len(vals), k))


if __name__ == "__main__":
    main(sys.argv[1:])
