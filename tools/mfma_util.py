"""Reduce tools/mfma_util.sh output: per grouped-conv kernel, MFMA busy cycles per launch (summed over the 1024 SIMDs)
against the kernel's duration -> fraction of the matrix pipes' time in use (SQ_VALU_MFMA_BUSY_CYCLES counts 32 cycles per
v_mfma_f32_16x16x4_f32, the 8-pass rate behind the 157 TFLOP/s fp32 peak)."""
import collections
import csv
import glob

CLK_GHZ = 2.4
SIMDS = 256 * 4
SHAPES = {"2d": "2D 32^2 C16 H64 B8", "3d": "3D 16^3 C16 H16 B8", "8c64": "3D 8^3 64->64 per group, 16 groups, B8",
          "4c64": "3D 4^3 64->64 per group, 16 groups, B8", "2c64": "3D 2^3 64->64 per group, 16 groups, B8",
          "8c64_2d": "2D 8^2 64->64 per group, 16 groups, B8", "8c32": "3D 8^3 C32 (32->32 per group), 16 groups, B8",
          "32c4": "3D 32^3 C4 (4->4 per group), 16 groups, B8"}


def main():
    lines = ["MFMA utilisation of the grouped-conv kernels (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA; durations from a",
             "separate --kernel-trace --stats run of the same command, tools/gconv_prof.py; clock taken as %.1f GHz)" % CLK_GHZ, ""]
    for cfg, shape in SHAPES.items():
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(f"gpurun_out/mfma_pmc_{cfg}/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur = {}
        for f in glob.glob(f"gpurun_out/mfma_stats_{cfg}/*kernel_stats.csv"):
            for r in csv.DictReader(open(f)):
                dur[r["Name"]] = float(r["AverageNs"])
        lines.append(shape)
        for k in sorted(acc):
            if "gconv" not in k or "reduce" in k or "bias" in k:
                continue
            busy = sum(acc[k]["SQ_VALU_MFMA_BUSY_CYCLES"]) / max(1, len(acc[k]["SQ_VALU_MFMA_BUSY_CYCLES"]))
            n = sum(acc[k]["SQ_INSTS_MFMA"]) / max(1, len(acc[k]["SQ_INSTS_MFMA"]))
            d = dur.get(k)
            if not d:
                continue
            util = busy / (d * CLK_GHZ * SIMDS)
            short = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            lines.append(f"  {short:42s} {d / 1e3:7.1f} us/launch  {n:12.0f} MFMA wave-instr  busy {busy:.3e} cyc  -> {100 * util:4.1f} % of the matrix pipes")
        lines.append("")
    import sys
    open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gconv_mfma_util.txt", "w").write("\n".join(lines))
    print("\n".join(lines))


if __name__ == "__main__":
    main()
