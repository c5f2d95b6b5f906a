#!/usr/bin/env python3
"""Readable lines out of a rocprofv3 *_kernel_stats.csv: name (up to the argument list), calls, average and minimum in us."""
import csv
import sys

for r in csv.DictReader(open(sys.argv[1])):
    print(f'{r["Name"].split("(")[0]:34s} calls {r["Calls"]:>4s} avg_us {float(r["AverageNs"]) / 1e3:9.2f} min_us {float(r["MinNs"]) / 1e3:9.2f}')
