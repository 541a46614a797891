#!/usr/bin/env python3
"""Turn rocprofv3 output into the committed profile summaries.

  tools/pmc_traffic.py <kernel_stats.csv> <fetch_counter.csv> <write_counter.csv> <out_prefix>

* copies the --kernel-trace --stats table (our kernels only) to <out_prefix>_kernel_stats.csv
* reduces the two PMC passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, collected in
  SEPARATE runs: TCC slots do not fit both) to HBM bytes per launch per kernel:
      bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of the
  bytes of a wide coalesced read (MI355X_MICROARCH.md §HBM), hence the factor 2.
  -> <out_prefix>_traffic.json and profiles/traffic_latest.json
"""
import csv
import json
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def pmc_avg(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            k = short(row["Kernel_Name"])
            acc[k][0] += float(row["Counter_Value"])
            acc[k][1] += 1
    return {k: v[0] / v[1] for k, v in acc.items() if v[1]}


def main():
    stats, fetch, write, prefix = sys.argv[1:5]
    ours = ("scatter_", "quad_kernel", "gather_", "splat_max_bwd", "slice_bwd_fused", "slice_bwd_sorted", "plane_sort", "positions_", "nn_kernel", "emd_", "occupancy")
    rows = []
    with open(stats) as f:
        for row in csv.DictReader(f):
            n = short(row["Name"])
            if any(o in n for o in ours):
                rows.append({"Name": n, "Calls": row["Calls"], "AverageNs": row["AverageNs"],
                             "MinNs": row["MinNs"], "MaxNs": row["MaxNs"], "TotalDurationNs": row["TotalDurationNs"]})
    with open(prefix + "_kernel_stats.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    fe, wr = pmc_avg(fetch, "FETCH_SIZE"), pmc_avg(write, "WRITE_SIZE")
    out = {"unit": "bytes per launch", "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE halves wide reads)",
           "kernels": {}}
    for k in sorted(set(fe) | set(wr)):
        if not any(o in k for o in ours):
            continue
        f_kib, w_kib = fe.get(k, 0.0), wr.get(k, 0.0)
        out["kernels"][k] = {"FETCH_SIZE_KiB": f_kib, "WRITE_SIZE_KiB": w_kib,
                             "hbm_bytes_per_launch": (2 * f_kib + w_kib) * 1024}
    for path in (prefix + "_traffic.json", os.path.join(os.path.dirname(prefix) or ".", "traffic_latest.json")):
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
