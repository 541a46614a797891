"""Reduce the rocprofv3 passes over tools/zoo_prof.py (tools/zoo_prof.sh) into one table per zoo shape:
per kernel the median duration (kernel trace), HBM bytes per launch (2*FETCH_SIZE + WRITE_SIZE, KiB -> bytes; separate
passes, gfx950 correction of MI355X_MICROARCH.md §HBM) and the SQ counters per launch (LDS conflict share, vector / LDS
instructions, wait share).  The dispatch sequence of every pass is cut at the marker launches (occupancy kernel).

    python3 tools/zoo_prof_report.py gpurun_out/zoo_prof [B8N4096 B2N16384]
"""
import collections
import csv
import glob
import statistics
import sys

sys.path.insert(0, "tools")
from zoo_prof import SHAPES, CONFIGS      # noqa: E402

MARK = "occupancy"
OURS = ("scatter", "quad_kernel", "gather", "splat_", "slice_", "sum_parts", "zero_slots", "add_inplace", "mhct_")


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def segments(path, value_of):
    """[{kernel: {counter: [values per launch]}}] per marker-delimited segment of one pass"""
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: (int(r["Dispatch_Id"]), r.get("Counter_Name", "")))
    segs = []
    cur = None
    last_marker = None
    for r in rows:
        k = short(r["Kernel_Name"])
        if MARK in k:
            if last_marker != r["Dispatch_Id"]:
                cur = collections.defaultdict(lambda: collections.defaultdict(list))
                segs.append(cur)
                last_marker = r["Dispatch_Id"]
            continue
        if cur is None or k.startswith("at::") or "rocclr" in k or "elementwise" in k or "distribution" in k:
            continue
        name, v = value_of(r)
        cur[k][name].append(v)
    return segs


def one(pattern):
    f = glob.glob(pattern, recursive=True)
    return f[0] if f else None


def main():
    root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/zoo_prof"
    names = [a for a in sys.argv[2:] if a in CONFIGS] or ["B8N4096", "B2N16384"]
    order = [(n, s) for n in names for s in SHAPES]
    passes = {}
    f = one(root + "/trace/**/*kernel_trace.csv")
    if f:
        passes["us"] = segments(f, lambda r: ("us", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    for tag in ("sq_a", "sq_b", "fetch", "write"):
        f = one(root + "/" + tag + "/**/*counter_collection.csv")
        if f:
            passes[tag] = segments(f, lambda r: (r["Counter_Name"], float(r["Counter_Value"])))
    for i, (cfg, (C, W, dim)) in enumerate(order):
        B, N = CONFIGS[cfg]
        P = B * N * 16
        G = W ** dim
        alg = 6 * 4 * dim * P + 5 * 4 * C * P + 6 * 4 * C * G * B * 16
        print(f"== {cfg} H16 C{C} W{W} {dim}D   algorithmic bytes per step {alg / 1e6:.1f} MB")
        kernels = []
        for p in passes.values():
            if i < len(p):
                for k in p[i]:
                    if k not in kernels:
                        kernels.append(k)
        tot_us, tot_bytes = 0.0, 0.0
        for k in kernels:
            if not any(o in k for o in OURS):
                continue
            def avg(tag, c):
                p = passes.get(tag)
                if not p or i >= len(p) or c not in p[i].get(k, {}):
                    return None
                v = p[i][k][c]
                return statistics.median(v) if tag == "us" else sum(v) / len(v)
            us = avg("us", "us")
            n_launch = len(passes["us"][i][k]["us"]) if "us" in passes and i < len(passes["us"]) and k in passes["us"][i] else 0
            fe, wr = avg("fetch", "FETCH_SIZE"), avg("write", "WRITE_SIZE")
            hbm = (2 * fe + wr) * 1024 if fe is not None and wr is not None else None
            conf, act = avg("sq_b", "SQ_LDS_BANK_CONFLICT"), avg("sq_b", "SQ_LDS_IDX_ACTIVE")
            valu, lds, waves = avg("sq_b", "SQ_INSTS_VALU"), avg("sq_b", "SQ_INSTS_LDS"), avg("sq_b", "SQ_WAVES")
            wc, wa, busy = avg("sq_a", "SQ_WAVE_CYCLES"), avg("sq_a", "SQ_WAIT_ANY"), avg("sq_a", "SQ_BUSY_CYCLES")
            per_step = n_launch / 12.0 if n_launch else 1.0        # launches of this kernel per step (zoo_prof.ITERS = 12)
            if us is not None:
                tot_us += us * per_step
            if hbm is not None:
                tot_bytes += hbm * per_step
            fmt = lambda v, s="%.3g": "-" if v is None else s % v
            print(f"  {k[:70]:70s} x{per_step:3.1f} {fmt(us, '%7.1f')} us  hbm {fmt(None if hbm is None else hbm / 1e6, '%7.1f')} MB  "
                  f"lds conflict/active {fmt(None if not act else conf / act, '%.2f')} (active {fmt(act)})  valu {fmt(valu)}  lds {fmt(lds)}  "
                  f"waves {fmt(waves)}  wait/wave-cyc {fmt(None if not wc else wa / wc, '%.2f')}")
        if tot_us:
            print(f"  -> step {tot_us:.1f} us, {alg / tot_us / 1e6 / 8:.3f} of 8 TB/s on the algorithmic bytes; measured HBM traffic "
                  f"{tot_bytes / 1e6:.1f} MB = {tot_bytes / alg:.2f}x")


if __name__ == "__main__":
    main()
