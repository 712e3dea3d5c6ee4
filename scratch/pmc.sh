#!/bin/bash
# scratch/pmc.sh <tag> <counters...>  : one rocprofv3 --pmc pass over a short bench run
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$tag.json 2> gpurun_out/pmc_$tag.err || { tail -5 gpurun_out/pmc_$tag.err; exit 1; }
python3 - <<PY
import csv,glob,collections,os
f=glob.glob('gpurun_out/pmc_$tag/**/*counter_collection.csv',recursive=True)[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if os.environ.get("KNAME","k_traverse") in r["Kernel_Name"]:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items(): print(k, sum(v)/len(v), len(v))
PY
