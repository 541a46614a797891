"""Where a model's training step spends its GPU time: reduce a rocprofv3 --kernel-trace --stats table (kernel_stats.csv of one
of the tools/*_step_bench.py runs) to kernel families — library GEMMs, this package's norm / grouped-conv / raster / lattice /
loss kernels, torch's elementwise and reduction kernels, MIOpen, copies — with each family's share of the kernel time and
its five largest kernels.      python3 tools/model_prof_report.py <kernel_stats.csv> [title]"""
import collections
import csv
import sys

FAMILIES = [
    ("library GEMM (rocBLAS / Tensile)", ("Cijk_", "gemm", "Gemm")),
    ("norms (ct_bnorm / ct_adain)", ("bn_", "adain_")),
    ("grouped conv (ct_gconv)", ("gconv",)),
    ("raster: Splat / Slice / fused core (ct_raster, ct_mhct)", ("mhct_core", "scatter", "gather", "slice_", "splat_", "quad_kernel", "sum_parts",
                                                                  "occupancy", "positions_", "zero_slots", "add_inplace")),
    ("lattice / so3 (ct_lattice)", ("lattice", "so3")),
    ("losses (ct_emd / ct_chamfer)", ("emd_", "bid_", "nn_kernel", "chamfer")),
    ("MIOpen", ("miopen", "MIOpen", "Conv", "naive_conv", "Im2Col", "transpose_")),
    ("copies / fills (runtime)", ("rocclr", "fillBuffer", "copyBuffer")),
    ("torch elementwise / reductions / optimizer", ("at::native", "elementwise", "reduce_kernel", "multi_tensor", "vectorized", "softmax", "nll_loss", "cat")),
]


def family(name):
    for fam, keys in FAMILIES:
        if any(k in name for k in keys):
            return fam
    return "other"


def main():
    path = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else path
    tot = collections.defaultdict(float)
    top = collections.defaultdict(list)
    min_calls = int(sys.argv[3]) if len(sys.argv) > 3 else 14      # the runs take >= 20 steps: rarer kernels are one-off
    dropped = 0.0
    for r in csv.DictReader(open(path)):
        ns = float(r["TotalDurationNs"])
        if int(r["Calls"]) < min_calls or "naive_conv" in r["Name"]:     # library auto-tuning (MIOpen find mode runs its naive kernels), set-up
            dropped += ns
            continue
        fam = family(r["Name"])
        tot[fam] += ns
        top[fam].append((ns, int(r["Calls"]), r["Name"]))
    total = sum(tot.values())
    print("== %s: %.1f ms of steady-state kernel time in the profiled run (%.1f ms of one-off kernels — library auto-tuning, set-up — left out)"
          % (title, total / 1e6, dropped / 1e6))
    for fam, ns in sorted(tot.items(), key=lambda kv: -kv[1]):
        print("  %5.1f %%  %s" % (100 * ns / total, fam))
        for kns, calls, name in sorted(top[fam], reverse=True)[:5]:
            short = name.replace("(anonymous namespace)::", "").replace("void ", "")[:110]
            print("            %5.1f %%  %6d calls  %s" % (100 * kns / total, calls, short))


if __name__ == "__main__":
    main()
