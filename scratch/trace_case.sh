#!/bin/bash
# Kernel trace of one bench_configs case: per-kernel totals and the per-launch durations of the frame's kernels.
# usage: scratch/trace_case.sh <tag> <case> [env assignments are inherited]
tag=$1; shift
out=gpurun_out/trace_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 scratch/bench_configs.py "$@" > $out/bench.json 2> $out/bench.err || exit 1
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
python3 - <<PY
import csv,glob,re
f=glob.glob('$out/trace/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
def short(k):
    m=re.search(r'(k_[a-z_]+)',k); return m.group(1) if m else k[:30]
# last third of the launches = the timed frame (two warm-ups before it)
n=len(rows); last=rows[2*n//3:]
t0=int(last[0]['Start_Timestamp'])
with open('$out/launches.txt','w') as o:
    for r in last:
        o.write('%10.1f us  +%8.1f us  %-18s grid %s\n' % ((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,short(r['Kernel_Name']),r.get('Grid_Size','')))
print(open('$out/launches.txt').read()[:6000])
PY
cat $out/bench.json | tail -3
