#!/bin/bash
# Round-2 profile set for the current build -> gpurun_out/prof_$1/ (copy what is to be judged into profiles/ with
# scratch/profile_r2_collect.py afterwards).  GPU_MAX_HW_QUEUES is set in the SHELL: under rocprofv3 the profiler's
# library initialises HIP before bench.py can set it.
tag=$1
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16
B="python3 bench.py --no-cpu-baseline --no-configs"
# 1. kernel traces of the headline: one frame at a time (the roofline's kernel time) and 8 in flight (the headline setting)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace1 -- $B --steps 20 --warmup 2 --frames-in-flight 1 > $out/bench_1inflight.json 2> $out/bench_1inflight.err || exit 1
cp $(find $out/trace1 -name "*kernel_stats.csv" | head -1) $out/c2_kernel_stats_1inflight.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace8 -- $B > $out/bench_default.json 2> $out/bench_default.err || exit 1
cp $(find $out/trace8 -name "*kernel_stats.csv" | head -1) $out/c2_kernel_stats_default.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/traceb -- python3 scratch/batch_launches.py 16 > $out/batch16.txt 2> $out/batch16.err || exit 1
cp $(find $out/traceb -name "*kernel_stats.csv" | head -1) $out/c2_kernel_stats_batch16.csv
# 2. PMC passes over the headline, one frame at a time (each pass its own run, within the per-block counter limits)
pass() { name=$1; shift; timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_$name -- $B --steps 2 --warmup 1 --frames-in-flight 1 > /dev/null 2> $out/pmc_$name.err || return 1; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY || exit 1
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 || exit 1
pass tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum || exit 1
pass tcc2 TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum || exit 1
pass write WRITE_SIZE TCC_EA0_WRREQ_64B_sum || exit 1
pass grbm GRBM_GUI_ACTIVE GRBM_TA_BUSY || exit 1
# 2b. the same passes over launches of 16 frames (the dominant launch of the default bench run)
bpass() { name=$1; shift; timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d $out/pmcb_$name -- python3 scratch/batch_launches.py 16 > /dev/null 2> $out/pmcb_$name.err || return 1; }
bpass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY || exit 1
bpass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 || exit 1
bpass tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum || exit 1
bpass tcc2 TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum || exit 1
bpass write WRITE_SIZE TCC_EA0_WRREQ_64B_sum || exit 1
# 3. kernel traces of the other BASELINE configs (scratch/bench_configs.py renders each fixture: 1 counting + 2 warm-up + 3 timed frames)
for c in c1_default c1_shadows_fresnel st_dof c4_combined st_motion_blur; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/cfg_$c -- python3 scratch/bench_configs.py $c > $out/cfg_$c.json 2> $out/cfg_$c.err || exit 1
  cp $(find $out/cfg_$c -name "*kernel_stats.csv" | head -1) $out/cfg_${c}_kernel_stats.csv
done
# 4. PMC of the shading-heavy frame (C3 at 8 spp to keep the passes short)
cpass() { name=$1; shift; RT_BENCH_SPP=8 timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d $out/c3pmc_$name -- python3 scratch/bench_configs.py st_dof > /dev/null 2> $out/c3pmc_$name.err || return 1; }
cpass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY || exit 1
cpass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 || exit 1
cpass tcc TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_EA0_ATOMIC_sum TCC_REQ_sum || exit 1
cpass grbm GRBM_GUI_ACTIVE GRBM_TA_BUSY || exit 1
rm -rf $out/trace1 $out/trace8 $out/traceb
# keep what the collection step reads (counter_collection.csv of the PMC passes, the copied kernel stats): the traces are large
for c in c1_default c1_shadows_fresnel st_dof c4_combined st_motion_blur; do rm -rf $out/cfg_$c; done
# digest on the box (the per-dispatch counter files are tens of MiB; gpurun brings back at most 64 MiB), then drop the raw files
RT_PROFILE_OUT=$out/digest python3 scratch/profile_r2_collect.py $tag > $out/collect.log 2>&1 || exit 1
rm -rf $out/pmc_*/ $out/pmcb_*/ $out/c3pmc_*/
du -sh $out
echo "profile set $tag done"
