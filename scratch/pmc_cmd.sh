#!/bin/bash
# scratch/pmc_cmd.sh <tag> <kernel substring> "<python script and args>" <counters...>: one --pmc pass, totals over all launches
tag=$1; kname=$2; cmd=$3; shift 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$tag -- python3 $cmd > gpurun_out/pmc_$tag.out 2> gpurun_out/pmc_$tag.err || { tail -5 gpurun_out/pmc_$tag.err; exit 1; }
KNAME=$kname python3 - <<PY
import csv,glob,collections,os
f=glob.glob('gpurun_out/pmc_$tag/**/*counter_collection.csv',recursive=True)[0]
acc=collections.defaultdict(float); n=collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    if os.environ["KNAME"] in r["Kernel_Name"]:
        acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
for k,v in sorted(acc.items()): print("%-40s total %.5g over %d launches" % (k, v, n[k]))
PY
