#!/bin/bash
# scratch/trace_levels.sh <case>: kernel timeline of the last frame of scratch/bench_configs.py <case> (start offset, duration, gap)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
c=${1:-c4_combined}
rm -rf /tmp/trl
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/trl -- python3 scratch/bench_configs.py $c > /dev/null 2>/tmp/trl.err || { tail -3 /tmp/trl.err; exit 1; }
python3 - <<'PY'
import csv, glob, re
f = glob.glob("/tmp/trl/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def kn(n):
    m = re.search(r"(k_[a-z_]+)", n); return m.group(1) if m else n[:20]
# the last frame: from the last k_raygen on
last = max(i for i, r in enumerate(rows) if "k_raygen" in r["Kernel_Name"])
fr = rows[last:]
t0 = int(fr[0]["Start_Timestamp"]); prev_end = t0
tot = 0
for r in fr:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us  %-18s dur %7.1f us  gap %6.1f us  grid %s" % ((s - t0) / 1e3, kn(r["Kernel_Name"]), (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Grid_Size", "")))
    prev_end = max(prev_end, e); tot += e - s
print("frame: %.1f us from first start to last end, %.1f us of kernels" % ((prev_end - t0) / 1e3, tot / 1e3))
PY
