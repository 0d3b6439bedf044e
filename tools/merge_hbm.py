#!/usr/bin/env python3
"""Joins the FETCH_SIZE and the WRITE_SIZE counter summaries of one command (two rocprofv3 --pmc runs: the two counters do
not fit one pass beside each other's companions) into one table: kernel, dispatches, FETCH_SIZE_KB, WRITE_SIZE_KB.
usage: merge_hbm.py <fetch.csv> <write.csv> <out.csv>"""
import csv
import sys


def rows(path):
    with open(path, newline="") as fh:
        r = [x for x in csv.reader(l for l in fh if not l.startswith("#"))]
    return {x[0]: x[1:] for x in r[1:]}


def main():
    fetch, write, out = sys.argv[1:4]
    f, w = rows(fetch), rows(write)
    with open(out, "w", newline="") as fh:
        fh.write("# two runs of: rocprofv3 --pmc FETCH_SIZE (resp. WRITE_SIZE) -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline ; "
                 "KB summed over all dispatches of each kernel in one frame; raw counter values (on gfx950 FETCH_SIZE tallies wide "
                 "coalesced reads at half their bytes)\n")
        wr = csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC)
        fh.write("kernel,dispatches,FETCH_SIZE_KB,WRITE_SIZE_KB\n")
        for k in f:
            if k in w:
                wr.writerow([k, int(float(f[k][0])), int(float(f[k][1])), int(float(w[k][1]))])


if __name__ == "__main__":
    main()
