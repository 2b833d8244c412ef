#!/usr/bin/env python3
"""Prints 'name calls avg_us' for the kernels of a rocprofv3 kernel_stats.csv whose name contains one of the given substrings."""
import csv
import sys

path, pats = sys.argv[1], sys.argv[2:]
for row in csv.DictReader(open(path)):
    name = row["Name"]
    if not pats or any(p in name for p in pats):
        short = name.split("(")[0].replace("void samd::", "").replace("samd::", "")
        print("  %-48s calls %5s  avg %10.1f us" % (short[:48], row["Calls"], float(row["AverageNs"]) / 1e3))
