#!/usr/bin/env python3
"""Condense a rocprofv3 output directory (csv format) into a small text summary for profiles/.

  python tools/prof_summary.py <dir-with-*_kernel_trace.csv> [<pmc-dir> ...] > profiles/xyz.txt

Per kernel AND grid size (so that the levels of the V-cycle are told apart): calls, average /
min / max duration.  For PMC directories: mean counter value per kernel and grid size.
"""
import collections
import csv
import glob
import os
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return name if len(name) <= 90 else name[:87] + "..."


def trace(d):
    files = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)
    agg = collections.defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            dur = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
            agg[(short(row["Kernel_Name"]), int(row["Grid_Size"]) if "Grid_Size" in row else
                 int(row.get("Grid_Size_X", 0)))].append(dur)
    tot = sum(sum(v) for v in agg.values())
    print("# kernel trace: %s  (total kernel time %.3f ms)" % (d, tot / 1e6))
    print("%-92s %12s %7s %10s %10s %10s %6s" % ("kernel", "grid", "calls", "avg_us", "min_us", "max_us", "%"))
    for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("%-92s %12d %7d %10.2f %10.2f %10.2f %6.2f" % (k, g, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3,
                                                             max(v) / 1e3, 100.0 * sum(v) / tot))


def pmc(d):
    files = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            agg[(short(row["Kernel_Name"]), int(row["Grid_Size"]), row["Counter_Name"])].append(float(row["Counter_Value"]))
    print("# counters: %s" % d)
    print("%-92s %12s %-12s %7s %16s" % ("kernel", "grid", "counter", "calls", "mean_value"))
    for (k, g, c), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("%-92s %12d %-12s %7d %16.3f" % (k, g, c, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    for d in sys.argv[1:]:
        has_pmc = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
        (pmc if has_pmc else trace)(d)
        print()
