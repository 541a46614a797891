"""Where a model's training step spends its GPU time: reduce a rocprofv3 --kernel-trace table (kernel_trace.csv of one of the
tools/*_step_bench.py runs) to kernel families.  Only the LAST QUARTER of the run's dispatches is counted — the
tools end with HIP-graph replays of the step, so that window is steady state: the libraries' auto-tuning runs of the first
steps (MIOpen's find mode executes every candidate solver, its naive kernels included) and the eager warm-up are left out — — library GEMMs, this package's norm / grouped-conv / raster / lattice /
loss kernels, torch's elementwise and reduction kernels, MIOpen, copies — with each family's share of the kernel time and
its largest kernels.      python3 tools/model_prof_report.py <kernel_trace.csv> [title] [kernels per family = 5]"""
import collections
import csv
import sys

FAMILIES = [
    ("pointwise-conv GEMMs on the f16 matrix pipes (ct_pwgemm: GEMM, weight prep, amax, slab sum)", ("pw_gemm", "pw2_gemm", "pw_amax", "pw_reduce", "pw_transpose", "pw_prep")),
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


NTOP = 5


def family(name):
    for fam, keys in FAMILIES:
        if any(k in name for k in keys):
            return fam
    return "other"


def main():
    path = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else path
    global NTOP
    NTOP = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    tot = collections.defaultdict(float)
    top = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
    rows.sort()
    rows = rows[int(0.75 * len(rows)):]         # the last quarter of the DISPATCHES (the auto-tuning runs are few, long launches early on)
    t0, t1 = rows[0][0], rows[-1][1]
    for a, b, name in rows:
        fam = family(name)
        tot[fam] += b - a
        rec = top[fam][name]
        rec[0] += b - a
        rec[1] += 1
    total = sum(tot.values())
    print("== %s: %.1f ms of kernel time in the steady-state window (the last quarter of the dispatches, %.3f s of wall time)"
          % (title, total / 1e6, (t1 - t0) / 1e9))
    for fam, ns in sorted(tot.items(), key=lambda kv: -kv[1]):
        print("  %5.1f %%  %s" % (100 * ns / total, fam))
        for kns, calls, name in sorted(((v[0], v[1], k) for k, v in top[fam].items()), reverse=True)[:NTOP]:
            short = name.replace("(anonymous namespace)::", "").replace("void ", "")[:110]
            print("            %5.1f %%  %6d calls  %7.1f us each  %s" % (100 * kns / total, calls, kns / calls / 1e3, short))


if __name__ == "__main__":
    main()
