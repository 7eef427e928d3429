#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc counter_collection CSVs under a directory (developer tool)."""
import collections
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else "smk_k_"
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if pat in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        print(k)
        for c in sorted(acc[k]):
            v = acc[k][c]
            print("   %-28s n=%3d mean %.6g" % (c, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    main()
