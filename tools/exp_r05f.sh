#!/bin/bash
# round 5, call f: counters of the fused Allan kernel; do two streams of one process overlap?
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05f
mkdir -p $OUT
cd $ROOT
timeout 300 python tools/experiments/stream_overlap.py > $OUT/stream_overlap.log 2>&1; cat $OUT/stream_overlap.log
cd /tmp && export TMPDIR=/tmp
export WARM=3
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/pmc_sq -o allan -- python $ROOT/tools/bench_allan.py > $OUT/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --kernel-trace -d $OUT/pmc_lds -o allan -- python $ROOT/tools/bench_allan.py > $OUT/pmc_lds.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o allan -- python $ROOT/tools/bench_allan.py > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o allan -- python $ROOT/tools/bench_allan.py > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace -d $OUT/pmc_grbm -o allan -- python $ROOT/tools/bench_allan.py > $OUT/pmc_grbm.log 2>&1
du -sh $OUT
