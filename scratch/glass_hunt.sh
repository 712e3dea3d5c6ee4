#!/bin/bash
# scratch/glass_hunt.sh: kernel traces of "st_dof then own_glass_env" in one process, several times (hunting an intermittent slow frame)
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/glass_hunt; rm -rf $out; mkdir -p $out
for i in 1 2 3 4; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof$i -- python3 $GRAFT_REPO_ROOT/scratch/bench_configs.py st_dof own_glass_env > $out/run$i.log 2>&1 || exit 1
  grep -o '"case": "own_glass_env".*"wall_ms": [0-9.]*' $out/run$i.log
  f=$(find /tmp/prof$i -name "*kernel_stats.csv" | head -1)
  cp "$f" $out/run${i}_kernel_stats.csv
  g=$(find /tmp/prof$i -name "*kernel_trace.csv" | head -1)
  python3 - "$g" > $out/run${i}_tail.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last render = the measured glass_env frame: print the 40 longest dispatches of the last 2000
last = rows[-3000:]
last.sort(key=lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), reverse=True)
for r in last[:25]:
    print("%10.3f ms  %s  grid %s wg %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r["Kernel_Name"][:60], r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Workgroup_Size_X", r.get("Workgroup_Size"))))
PY
done
