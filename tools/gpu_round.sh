#!/bin/bash
# Run on the GPU box via gpurun: tests, smoke, bench, rocprofv3 kernel trace + PMC passes.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python __graft_entry__.py --smoke 2>&1 | tail -3
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 3000 $OUT/bench.json; tail -5 $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --steps 10 --warmup 2 --cpu-baseline-seconds 0"
rocprofv3 --kernel-trace --stats -d $OUT/prof_trace -o bench -- python $ROOT/bench.py --cpu-baseline-seconds 0 > $OUT/prof_trace.log 2>&1    # the default command: 200 + 10 steps
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/prof_pmc_write -o bench -- $B > $OUT/prof_pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/prof_pmc_fetch -o bench -- $B > $OUT/prof_pmc_fetch.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/prof_pmc_sq -o bench -- $B > $OUT/prof_pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace -d $OUT/prof_pmc_grbm -o bench -- $B > $OUT/prof_pmc_grbm.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/prof_allan -o bench -- python $ROOT/tools/bench_allan.py > $OUT/prof_allan.log 2>&1
ls $OUT
