#!/bin/bash
# PMC passes (one rocprofv3 run each) over one bench_configs case; per-kernel means -> gpurun_out/pmc_<tag>/summary.txt
# usage: scratch/pmc_case.sh <tag> <case>     (RT_BENCH_SPP overrides the sample count to keep the passes short)
tag=$1; shift
out=gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
pass() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 scratch/bench_configs.py $CASE > /dev/null 2> $out/$name.err || return 1; }
CASE="$@"
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY || exit 1
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 || exit 1
pass tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum || exit 1
pass tcc2 TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum TCC_WRITE_sum || exit 1
pass grbm GRBM_GUI_ACTIVE GRBM_TA_BUSY || exit 1
python3 - <<PY
import csv,glob,collections,re
acc=collections.defaultdict(lambda: collections.defaultdict(float))
cnt=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('$out/*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        m=re.search(r'(k_[a-z_]+)', r['Kernel_Name'])
        if not m: continue
        acc[m.group(1)][r['Counter_Name']]+=float(r['Counter_Value']); cnt[m.group(1)][r['Counter_Name']]+=1
with open('$out/summary.txt','w') as o:
    for k in acc:
        o.write('[%s] totals over all launches of the run (3 frames)\n' % k)
        for c,v in sorted(acc[k].items()): o.write('  %-28s %.6g   (%d launches)\n' % (c, v, cnt[k][c]))
print(open('$out/summary.txt').read())
PY
