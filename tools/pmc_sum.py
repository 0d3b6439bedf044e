"""Sums rocprofv3 --pmc counters per kernel.  usage: pmc_sum.py <dir with *_counter_collection.csv> [out.csv] [header comment]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    src = sys.argv[1]
    files = glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
    sums = defaultdict(lambda: defaultdict(float))
    dispatches = defaultdict(set)
    names = set()
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"].split("(")[0]
                sums[k][row["Counter_Name"]] += float(row["Counter_Value"])
                dispatches[k].add(row["Dispatch_Id"])
                names.add(row["Counter_Name"])
    names = sorted(names)
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    if len(sys.argv) > 3:
        out.write("# " + sys.argv[3] + "\n")
    out.write("kernel,dispatches," + ",".join(names) + "\n")
    for k in sorted(sums):
        out.write('"%s",%d,' % (k, len(dispatches[k])) + ",".join("%d" % sums[k][n] for n in names) + "\n")


if __name__ == "__main__":
    main()
