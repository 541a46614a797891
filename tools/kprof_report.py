"""Per-kernel summary of the passes of tools/kprof.sh: mean duration, launches, HBM bytes per launch (2 * FETCH_SIZE +
WRITE_SIZE, KiB -> bytes: the gfx950 correction of MI355X_MICROARCH.md), SQ counters per launch.
    python3 tools/kprof_report.py gpurun_out/kprof_TAG [min_us]"""
import collections
import csv
import glob
import sys


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def one(pattern):
    f = glob.glob(pattern, recursive=True)
    return f[0] if f else None


def main():
    root = sys.argv[1]
    min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
    dur = collections.defaultdict(list)
    f = one(root + "/trace/**/*kernel_trace.csv")
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    ctr = collections.defaultdict(lambda: collections.defaultdict(list))
    for tag in ("sq_a", "sq_b", "sq_c", "fetch", "write"):
        f = one(root + "/" + tag + "/**/*counter_collection.csv")
        if not f:
            continue
        for r in csv.DictReader(open(f)):
            ctr[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    rows = sorted(dur.items(), key=lambda kv: -sum(kv[1]))
    for k, d in rows:
        d2 = d[len(d) // 4:] if len(d) >= 8 else d              # drop warm-up launches
        us = sum(d2) / len(d2)
        if us < min_us or k.startswith("at::") or "rocclr" in k:
            continue
        c = {n: sum(v[len(v) // 4:]) / max(1, len(v[len(v) // 4:])) for n, v in ctr[k].items()}
        hbm = (2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024
        print(f"{k}\n    {us:8.1f} us x{len(d)}  hbm {hbm / 1e6:7.1f} MB ({hbm / us / 1e6:5.2f} TB/s)  fetch {2 * c.get('FETCH_SIZE', 0) * 1024 / 1e6:.1f} write {c.get('WRITE_SIZE', 0) * 1024 / 1e6:.1f} MB")
        cyc = us * 2.4e3
        g = c.get
        if g("SQ_WAVE_CYCLES"):
            print(f"    waves {g('SQ_WAVES', 0):.0f}  wave-cycles {g('SQ_WAVE_CYCLES'):.3g}  busy {g('SQ_BUSY_CYCLES', 0):.3g}  wait_any/wave-cyc {g('SQ_WAIT_ANY', 0) / g('SQ_WAVE_CYCLES'):.2f}  "
                  f"wait_inst_any {g('SQ_WAIT_INST_ANY', 0) / g('SQ_WAVE_CYCLES'):.2f}  wait_inst_lds {g('SQ_WAIT_INST_LDS', 0) / g('SQ_WAVE_CYCLES'):.2f}  active_inst_any {g('SQ_ACTIVE_INST_ANY', 0) / g('SQ_WAVE_CYCLES'):.2f}")
        print(f"    insts: valu {g('SQ_INSTS_VALU', 0):.3g} mfma {g('SQ_INSTS_MFMA', 0):.3g} lds {g('SQ_INSTS_LDS', 0):.3g} salu {g('SQ_INSTS_SALU', 0):.3g} vmem {g('SQ_INSTS_VMEM', 0):.3g}  "
              f"lds conflict/active {g('SQ_LDS_BANK_CONFLICT', 0) / max(1.0, g('SQ_LDS_IDX_ACTIVE', 0)):.2f} (active {g('SQ_LDS_IDX_ACTIVE', 0):.3g})")
        if g("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
            # busy cycles summed over SIMDs (4 per CU x 256 CUs); /4 as tools/mfma_util.py does
            print(f"    mfma busy {g('SQ_VALU_MFMA_BUSY_CYCLES', 0):.3g} cyc -> {100 * g('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 4 / (256 * cyc):.1f} % of the matrix pipes   vmem inst cycles {g('SQ_INST_CYCLES_VMEM', 0):.3g}  active_inst_vmem {g('SQ_ACTIVE_INST_VMEM', 0):.3g}")


if __name__ == "__main__":
    main()
