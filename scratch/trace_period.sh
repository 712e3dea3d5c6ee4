#!/bin/bash
# scratch/trace_period.sh <lib tag> <frames in flight>: kernel trace of bench.py, prints the period between k_raygen starts and the bench line
tag=$1; F=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tp_$tag
RT_MI355X_LIB=$GRAFT_REPO_ROOT/raytracer_amd/librt_mi355x_$tag.so timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tp_$tag -- python3 bench.py --steps 64 --warmup 1 --no-cpu-baseline --frames-in-flight $F > gpurun_out/tp_$tag.json 2>/dev/null || exit 1
python3 - <<PY
import csv,glob,json
f=glob.glob('gpurun_out/tp_$tag/*/*kernel_trace.csv')[0]
st=sorted(int(r['Start_Timestamp']) for r in csv.DictReader(open(f)) if 'k_raygen' in r['Kernel_Name'])
st=st[-60:]
tr=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(f)) if 'k_traverse' in r['Kernel_Name']][-60:]
d=json.load(open('gpurun_out/tp_$tag.json'))
print("$tag F=$F: trace period %.3f ms, k_traverse %.0f us; bench says %.3f ms/step kernel %.3f" % ((st[-1]-st[0])/1e6/(len(st)-1), sum(tr)/len(tr), d['ms_per_step'], d['roofline']['kernel_ms']))
PY
