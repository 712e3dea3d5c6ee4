#!/bin/bash
# scratch/pmc_kernel.sh <tag> <case>: a few PMC passes, per-kernel MEANS PER LAUNCH of the largest launches (level 0) only
tag=$1; shift
out=gpurun_out/pmck_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
pass() { name=$1; shift; timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 scratch/bench_configs.py $CASE > /dev/null 2> $out/$name.err || return 1; }
CASE="$@"
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY || exit 1
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS || exit 1
pass tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum || exit 1
pass tcc2 TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum TCC_WRITE_sum TCC_EA0_WRREQ_64B_sum || exit 1
pass grbm GRBM_GUI_ACTIVE GRBM_TA_BUSY || exit 1
python3 - <<PY
import csv,glob,collections,re
rows=collections.defaultdict(list)
for f in glob.glob('$out/*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        m=re.search(r'(k_[a-z_]+)', r['Kernel_Name'])
        if m: rows[(m.group(1), r['Counter_Name'])].append(float(r['Counter_Value']))
ks=sorted(set(k for k,_ in rows))
for k in ks:
    print('[%s] max over launches (the level-0 launch)' % k)
    for (kk,c),v in sorted(rows.items()):
        if kk==k: print('  %-28s %.6g' % (c, max(v)))
PY
