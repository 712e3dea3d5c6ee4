#!/bin/bash
# Round-3 profile set for the current build -> gpurun_out/prof_$1/digest (what scratch/profile_r3_collect.py made of it on the box;
# copy into profiles/ afterwards).  GPU_MAX_HW_QUEUES / RT_RAYS_PER_LANE are set in the SHELL: under rocprofv3 the profiler's
# library initialises HIP before bench.py can set them.  Every rocprofv3 run has the program itself after "--".
tag=$1
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16 RT_RAYS_PER_LANE=8
B="python3 bench.py --no-cpu-baseline --no-configs"
# 1. kernel traces of the headline: one frame per launch, the default run, nothing but launches of 16 frames
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace1 -- python3 scratch/batch_launches.py 1 > $out/batch1.txt 2> $out/batch1.err || exit 1
cp $(find $out/trace1 -name "*kernel_stats.csv" | head -1) $out/c2_kernel_stats_1inflight.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace8 -- $B > $out/bench_default.json 2> $out/bench_default.err || exit 1
cp $(find $out/trace8 -name "*kernel_stats.csv" | head -1) $out/c2_kernel_stats_default.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/traceb -- python3 scratch/batch_launches.py 16 > $out/batch16.txt 2> $out/batch16.err || exit 1
cp $(find $out/traceb -name "*kernel_stats.csv" | head -1) $out/c2_kernel_stats_batch16.csv
echo "traces done"
# 2. PMC passes over the headline (each pass its own run, within the per-block counter limits): one frame per launch, then launches of 16
pass() { name=$1; shift; timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_$name -- python3 scratch/batch_launches.py 1 > /dev/null 2> $out/pmc_$name.err || return 1; }
bpass() { name=$1; shift; timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d $out/pmcb_$name -- python3 scratch/batch_launches.py 16 > /dev/null 2> $out/pmcb_$name.err || return 1; }
for p in pass bpass; do
  $p sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY || exit 1
  $p sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 || exit 1
  $p tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum || exit 1
  $p tcc2 TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum || exit 1
  $p write WRITE_SIZE TCC_EA0_WRREQ_64B_sum || exit 1
  $p grbm GRBM_GUI_ACTIVE GRBM_TA_BUSY || exit 1
  echo "$p done"
done
# 3. the other BASELINE configs: kernel trace, then PMC passes (VALU, HBM reads, HBM writes), 8 spp for the many-sample frames
for c in c1_default c1_shadows_fresnel st_dof c4_combined st_motion_blur; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/cfg_$c -- python3 scratch/bench_configs.py $c > $out/cfg_$c.json 2> $out/cfg_$c.err || exit 1
  cp $(find $out/cfg_$c -name "*kernel_stats.csv" | head -1) $out/cfg_${c}_kernel_stats.csv
  cpass() { name=$1; shift; RT_BENCH_SPP=8 timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d $out/cfgpmc_${c}_$name -- python3 scratch/bench_configs.py $c > /dev/null 2> $out/cfgpmc_${c}_$name.err || return 1; }
  cpass sq SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAVES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 || exit 1
  cpass grbm GRBM_GUI_ACTIVE GRBM_TA_BUSY || exit 1
  cpass rd TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum || exit 1
  cpass wr WRITE_SIZE TCC_EA0_ATOMIC_sum || exit 1
  rm -rf $out/cfg_$c
  echo "config $c done"
done
rm -rf $out/trace1 $out/trace8 $out/traceb
# digest on the box (the per-dispatch counter files are tens of MiB; gpurun brings back at most 64 MiB), then drop the raw files
RT_PROFILE_OUT=$out/digest python3 scratch/profile_r3_collect.py $tag r03 > $out/collect.log 2>&1 || { tail -5 $out/collect.log; exit 1; }
rm -rf $out/pmc_*/ $out/pmcb_*/ $out/cfgpmc_*/
du -sh $out
echo "profile set $tag done"
