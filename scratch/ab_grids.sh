#!/bin/bash
# scratch/ab_grids.sh: grids of k_shade / k_shadow sized by what is resident (default) against 8 workgroups per CU
run() { echo "== shade ${1:-default} shadow ${2:-default}"; RT_SHADE_GRID_PER_CU=$1 RT_SHADOW_GRID_PER_CU=$2 timeout -k 10 400 python scratch/bench_configs.py c1_default c1_shadows_fresnel st_dof c4_combined st_motion_blur own_glass_env gal_shapes 2>&1 | grep config | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   %-22s %9.3f ms' % (d['case'], d['kernel_ms']))"; }
unset RT_SHADE_GRID_PER_CU RT_SHADOW_GRID_PER_CU
echo "== default (resident)"; timeout -k 10 400 python scratch/bench_configs.py c1_default c1_shadows_fresnel st_dof c4_combined st_motion_blur own_glass_env gal_shapes 2>&1 | grep config | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   %-22s %9.3f ms' % (d['case'], d['kernel_ms']))"
run 8 8
