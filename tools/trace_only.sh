ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/prof_trace
rocprofv3 --kernel-trace --stats -d $OUT/prof_trace -o bench -- python $ROOT/bench.py --cpu-baseline-seconds 0 > $OUT/prof_trace.log 2>&1
tail -c 600 $OUT/prof_trace.log | cut -c1-400
