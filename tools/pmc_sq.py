"""Reduce the SQ counter CSVs of tools/pmc_sq.sh: per kernel, the average per launch of each counter."""
import collections
import csv
import glob
import sys


def main():
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for f in glob.glob("gpurun_out/pmc_sq_*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
    for k in sorted(acc):
        if pat and pat not in k:
            continue
        print(k[:150])
        for c in sorted(acc[k]):
            n = len(launches[(k, c)])
            print(f"    {c:24s} {acc[k][c] / n:14.4e}   ({n} launches)")


if __name__ == "__main__":
    main()
