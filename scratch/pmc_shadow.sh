#!/bin/bash
# scratch/pmc_shadow.sh <case>: instruction classes and wait cycles of k_shadow for the default library and every variant library
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
c=${1:-st_motion_blur}
for lib in "" $(ls raytracer_amd/librt_mi355x_*.so 2>/dev/null); do
  echo "== ${lib:-default}"
  export RT_MI355X_LIB=${lib:+$PWD/$lib}
  for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES"; do
    rm -rf /tmp/pmcs
    RT_BENCH_SPP=8 timeout -s KILL 200 rocprofv3 --pmc $pass --output-format csv -d /tmp/pmcs -- python3 scratch/bench_configs.py $c > /dev/null 2>/tmp/pmcs.err || { tail -3 /tmp/pmcs.err; exit 1; }
    python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("/tmp/pmcs/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = "k_shadow" if "k_shadow" in k else "k_shade" if "k_shade" in k else "k_traverse_small" if "k_traverse_small" in k else None
        if k: acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k in acc:
    print("   %-18s " % k + "  ".join("%s %.4g" % (n.replace("SQ_", ""), v) for n, v in sorted(acc[k].items())))
PY
  done
done
