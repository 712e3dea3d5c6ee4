#!/bin/bash
# Full profile set for the current build: kernel trace stats + PMC passes (each its own run) -> gpurun_out/prof_$1/
tag=$1
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --frames-in-flight 1 > $out/bench.json 2> $out/bench.err || exit 1
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
pass() { name=$1; shift; timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_$name -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --frames-in-flight 1 > /dev/null 2> $out/pmc_$name.err || return 1; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY || exit 1
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 || exit 1
pass tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum || exit 1
pass tcc2 TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum || exit 1
pass fetch FETCH_SIZE || exit 1
pass write WRITE_SIZE TCC_EA0_WRREQ_64B_sum || exit 1
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum || exit 1
pass grbm GRBM_GUI_ACTIVE GRBM_TA_BUSY || exit 1
python3 - <<PY
import csv,glob,collections,json
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$out/pmc_*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        k='k_traverse' if 'k_traverse' in k else 'k_shade' if 'k_shade' in k else 'k_raygen' if 'k_raygen' in k else 'k_resolve' if 'k_resolve' in k else None
        if k: acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
with open('$out/pmc_summary.txt','w') as o:
    for k in acc:
        o.write('[%s] mean per launch over %d launches\n' % (k, len(next(iter(acc[k].values())))))
        for c,v in sorted(acc[k].items()): o.write('  %-36s %.6g\n' % (c, sum(v)/len(v)))
print(open('$out/pmc_summary.txt').read())
PY
