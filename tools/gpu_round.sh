#!/bin/bash
# Evidence run on the GPU box (via gpurun): GPU tests, smoke, the default bench, and the rocprofv3 passes whose summaries are
# committed under profiles/.  Everything of ONE invocation lands in gpurun_out/round_<tag>/ (raw rocprofv3 databases
# included), and tools/summarize_prof.py regenerates profiles/<tag>_* from exactly that directory -- so a summary can
# always be traced back to the raw run it came from, and every file carries the sha256 of the libginsim.so that ran.
#   usage: tools/gpu_round.sh <tag>        e.g. r02a   (one tag per box / invocation; all of them are kept)
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/round_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
sha256sum gnss-ins-sim_amd/lib/libginsim.so | cut -c1-16 > $OUT/lib_sha256.txt
(rocm-smi --showproductname 2>/dev/null | grep -i "card series\|GFX" | head -2; uname -n) > $OUT/box.txt 2>&1
python -X faulthandler -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
cp gpurun_out/parity_margins.json $OUT/ 2>/dev/null
python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err; python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("bench: %.4g %s, %.3f ms/step, kernel %.3f ms, frac %.3f, traffic %s" % (d["value"], d["unit"], d["ms_per_step"],
      d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"], d["roofline"]["traffic"]))
for l in d.get("configs", []):
    if "roofline" in l:
        print("  leg %-22s kernel %.3f ms  bound %s frac %s" % (l["name"], l["roofline"]["kernel_ms_avg"], l["roofline"]["bound"], l["roofline"]["frac"]))
PY
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --pmc-child"
# the default command (200 timed + 30 warm-up steps, all legs) under the kernel trace: per-kernel averages that must agree with the HIP events
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_trace -o bench -- python $ROOT/bench.py --cpu-baseline-seconds 0 --pmc off > $OUT/prof_trace.json 2> $OUT/prof_trace.log
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/prof_pmc_write -o bench -- $B > $OUT/prof_pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/prof_pmc_fetch -o bench -- $B > $OUT/prof_pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/prof_pmc_sq -o bench -- $B > $OUT/prof_pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --kernel-trace -d $OUT/prof_pmc_mix -o bench -- $B > $OUT/prof_pmc_mix.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace -d $OUT/prof_pmc_lds -o bench -- $B > $OUT/prof_pmc_lds.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace -d $OUT/prof_pmc_grbm -o bench -- $B > $OUT/prof_pmc_grbm.log 2>&1
# round 5: where the waves of the Allan kernels wait (SQ_WAIT_ANY next to the LDS side)
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --kernel-trace -d $OUT/prof_pmc_wait -o bench -- $B > $OUT/prof_pmc_wait.log 2>&1
# round 4: where the store-bound launches wait (SQ side of the vector-memory path)
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL --kernel-trace -d $OUT/prof_pmc_store -o bench -- $B > $OUT/prof_pmc_store.log 2>&1
# (TA_* / TCP_* / TCC_* stall counters: the rocprofv3 of this image aborts on a TA_* pass and then hangs in its signal handler --
# 37 minutes of a gpurun call in round 4 -- so no pass of those blocks; every pass below a timeout)
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_allan -o bench -- python $ROOT/tools/bench_allan.py > $OUT/prof_allan.json 2> $OUT/prof_allan.log
du -sh $OUT
# the summaries are made HERE (the raw databases do not fit the 64 MiB gpurun copies back): profiles/<tag>_* of this box's copy of the
# repository -> gpurun_out/evidence_<tag>/; of the raw run the logs, the JSON lines and the kernel-trace database of the bench
# command stay (what a reader needs to re-derive <tag>_kernel_trace_stats.csv), the counter databases go
cd $ROOT
python tools/summarize_prof.py $TAG --traffic > $OUT/summarize.log 2>&1; tail -2 $OUT/summarize.log
mkdir -p gpurun_out/evidence_$TAG && cp profiles/${TAG}_* profiles/pmc_traffic.json gpurun_out/evidence_$TAG/ 2>/dev/null
find $OUT -name "*.db" ! -path "*prof_trace*" -delete
find $OUT -name "*.db" -size +45M -delete
du -sh $OUT gpurun_out/evidence_$TAG
