#!/bin/bash
cd "$(dirname "$0")/../.."
R=$PWD
mkdir -p gpurun_out
CLOUDCT_LIB=$R/cloud_transformers_amd/lib/libcloudct_s3st.so python tools/dev/s3_stamps.py > gpurun_out/r6_s3_stamps.txt 2>&1
CLOUDCT_LIB=$R/cloud_transformers_amd/lib/libcloudct_s3st.so python tools/dev/s3_stamps.py 32 8 8 2048 >> gpurun_out/r6_s3_stamps.txt 2>&1
cat gpurun_out/r6_s3_stamps.txt
OUT=$R/gpurun_out/s3_prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/tools/dev/zoo_one.py 32 8 3 8 4096 > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
  --output-format csv -d $OUT/sq_a -o a -- python3 $R/tools/dev/zoo_one.py 32 8 3 8 4096 > $OUT/sq_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVES \
  --output-format csv -d $OUT/sq_b -o b -- python3 $R/tools/dev/zoo_one.py 32 8 3 8 4096 > $OUT/sq_b.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
for tag in ("sq_a", "sq_b"):
    f = glob.glob("gpurun_out/s3_prof/%s/**/*counter_collection.csv" % tag, recursive=True)
    if not f: print("no", tag); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0][:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_INSTS_VALU"): n[k] += 1
    for k in acc:
        print(tag, k, {c: "%.3g" % (v / max(1, n[k])) for c, v in acc[k].items()})
for f in glob.glob("gpurun_out/s3_prof/trace/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:1500])
PY
