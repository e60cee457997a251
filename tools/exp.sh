#!/bin/bash
# scratch experiment driver for one gpurun call (edited per call; results under gpurun_out/exp_<tag>/)
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/exp_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $ROOT
L=$ROOT/gnss-ins-sim_amd/lib
for rep in 1 2 3; do
  for t in ginsim ginsim_ntab; do GINSIM_LIB=$L/lib$t.so timeout 300 python tools/experiments/ab_kernels.py >> $OUT/ab.jsonl 2>> $OUT/ab.err; done
done
python - <<PY
import json
rows = [json.loads(l) for l in open("$OUT/ab.jsonl") if l.startswith('{')]
for tag in ('c3_ps', 'c3_end', 'f32_nothing', 'f32_kept', 'f64_nothing', 'f64_kept'):
    for lib in sorted({r['lib'] for r in rows}):
        v = [r[tag] for r in rows if r['lib'] == lib]
        print('%-12s %-22s min %s  avg %s  sha %s' % (tag, lib, [x['ms_min'] for x in v], [x['ms_avg'] for x in v], {x['end_sha'] for x in v}))
PY
cd /tmp && export TMPDIR=/tmp
for t in ginsim ginsim_ntab; do
  GINSIM_LIB=$L/lib$t.so REPS=2 timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES --kernel-trace -d $OUT/pmc_$t -o p -- python $ROOT/tools/experiments/ab_kernels.py > $OUT/pmc_$t.log 2>&1
done
cd $ROOT
python - <<PY
import sqlite3, glob
for t in ('ginsim', 'ginsim_ntab'):
    dbs = glob.glob("$OUT/pmc_%s/**/*.db" % t, recursive=True)
    if not dbs: print('no db', t); continue
    con = sqlite3.connect(dbs[0])
    print(t)
    rows = {}
    for k, c, v in con.execute("select kernel_name, counter_name, avg(value) from counters_collection where kernel_name like '%mc_kernel%' group by kernel_name, counter_name"):
        rows.setdefault(k[:70], {})[c] = v
    for k, c in rows.items():
        print('   %-70s conflict/active %.3f  lds insts %.3g  wait_lds %.3g  valu %.4g  wave_cycles %.4g' % (k, c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1), c['SQ_INSTS_LDS'], c['SQ_WAIT_INST_LDS'], c['SQ_INSTS_VALU'], c['SQ_WAVE_CYCLES']))
PY
