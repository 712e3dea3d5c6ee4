#!/bin/bash
# scratch/pmc_fp32.sh: one rocprofv3 --pmc pass over launches of 16 C2 frames, parity kernel and FP32 variant -> gpurun_out/fp32_pmc.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
for v in parity fp32; do
  rm -rf /tmp/pmcf_$v
  if [ $v = fp32 ]; then export RT_BATCH_SCHEDULE=fp32; fi
  timeout -s KILL 180 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmcf_$v -- python3 scratch/batch_launches.py 16 > /dev/null 2>&1 || exit 1
done
python3 - <<'PY' | tee gpurun_out/fp32_pmc.txt
import csv, glob, collections
print("# rocprofv3 --pmc, one pass each, scratch/batch_launches.py 16 (launches of 16 C2 frames): k_traverse per launch")
for v in ("parity", "fp32"):
    acc = collections.defaultdict(float); n = 0
    for f in glob.glob("/tmp/pmcf_%s/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_traverse" in r["Kernel_Name"] and "small" not in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Counter_Name"] == "SQ_INSTS_VALU": n += 1
    i = acc["SQ_INSTS_VALU"] / max(1, n)
    f64 = (acc["SQ_INSTS_VALU_FMA_F64"] + acc["SQ_INSTS_VALU_MUL_F64"] + acc["SQ_INSTS_VALU_ADD_F64"] + acc["SQ_INSTS_VALU_TRANS_F64"]) / max(1, n)
    busy = acc["GRBM_GUI_ACTIVE"] / max(1, n) / 2.4e9 / 8   # (summed over the 8 XCDs)
    print("%-7s launches %d  VALU wave-instructions %.4g  of which FP64 fma/mul/add/trans %.3g (%.1f %%)  active lanes %.1f %%  busy %.2f ms  issue %.1f %%"
          % (v, n, i, f64, 100 * f64 / i, 100 * acc["SQ_THREAD_CYCLES_VALU"] / 64 / acc["SQ_INSTS_VALU"], busy * 1e3, 100 * i * 4 / (1024 * 2.4e9) / busy))
PY
