#!/bin/bash
# scratch/ab_traverse.sh: the default library and every variant library on launches of 16 C2 frames: kernel time (HIP events),
# node fetches and sphere tests per ray (counting pass of bench.py), VALU instructions and active lanes (one rocprofv3 --pmc pass each)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16 RT_RAYS_PER_LANE=8
for lib in "" $(ls raytracer_amd/librt_mi355x_*.so 2>/dev/null); do
  echo "== ${lib:-default}"
  export RT_MI355X_LIB=${lib:+$PWD/$lib}
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-configs 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; r=d['roofline']
print('   bench: %.1f Mrays/s  %.4f ms/frame  solo %.3f ms  kernel(16 frames) %.3f ms  kernel(1 frame) %.3f ms  fetches/ray %.2f  sphere tests/ray %.2f  box tests/ray %.1f' % (d['value'], d['ms_per_frame'], d['ms_per_frame_solo'], r['kernel_ms'], r['kernel_ms_one_frame_per_launch'], c['node_fetches_per_ray'], c['prim_tests_per_ray'], c['box_tests_per_ray']))" || exit 1
  rm -rf /tmp/pmcab
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmcab -- python3 scratch/batch_launches.py 16 > /dev/null 2>&1 || exit 1
  python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/pmcab/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_traverse" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
rays = 16 * 1920 * 1080
busy = m["GRBM_GUI_ACTIVE"] / 8 / 2.4e9
print("   pmc per launch of 16 frames: VALU %.4g wave-instr (%.1f per ray-wave slot, %.0f lane-instr per ray)  lanes %.1f %%  issue %.1f %%  busy %.3f ms  LDS instr %.3g  VMEM rd %.3g  waiting %.1f %%"
      % (m["SQ_INSTS_VALU"], m["SQ_INSTS_VALU"] * 64 / rays, m["SQ_THREAD_CYCLES_VALU"] / rays, 100 * m["SQ_THREAD_CYCLES_VALU"] / 64 / m["SQ_INSTS_VALU"],
         100 * m["SQ_INSTS_VALU"] * 4 / (1024 * 2.4e9) / busy, busy * 1e3, m["SQ_INSTS_LDS"], m["SQ_INSTS_VMEM_RD"], 100 * m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]))
PY
done
