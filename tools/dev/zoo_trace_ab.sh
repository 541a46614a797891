# kernel lists of one zoo shape with and without the arrival tickets: bash tools/dev/zoo_trace_ab.sh C W dim B N
R=/root/repo; O=$R/gpurun_out/zta; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/a -o a -- python3 $R/tools/dev/zoo_one_pass.py "$@" > $O/a.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/b -o b -- python3 $R/tools/dev/zoo_one_pass.py "$@" --no-tickets > $O/b.log 2>&1
cd $R
echo "== tickets"; python3 tools/dev/trace_short.py $(find $O/a -name "*kernel_trace.csv") 12
echo "== plain"; python3 tools/dev/trace_short.py $(find $O/b -name "*kernel_trace.csv") 12
