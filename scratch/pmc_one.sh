#!/bin/bash
# scratch/pmc_one.sh <case> [spp]: VALU issue and lane fractions per kernel of one bench_configs case (one rocprofv3 --pmc pass)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
c=$1; spp=${2:-4}
rm -rf /tmp/pmc1
RT_BENCH_SPP=$spp timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc1 -- python3 scratch/bench_configs.py $c > /dev/null 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("/tmp/pmc1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z_]+)", r["Kernel_Name"])
        if m:
            acc[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    i = v.get("SQ_INSTS_VALU", 0)
    if i < 1e6: continue
    busy = v["GRBM_GUI_ACTIVE"] / 8 / 2.4e9
    print("%-18s busy %8.2f ms  issue %5.1f %%  lanes %5.1f %%  waves waiting %4.1f %%" % (k, busy * 1e3, 100 * i * 4 / (1024 * 2.4e9) / busy, 100 * v["SQ_THREAD_CYCLES_VALU"] / 64 / i, 100 * v["SQ_WAIT_ANY"] / max(1, v["SQ_WAVE_CYCLES"])))
PY
