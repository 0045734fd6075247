#!/bin/bash
# the wave / group knobs again on the build whose staging got cheaper (second half of round 4)
mkdir -p gpurun_out/r04
out=gpurun_out/r04/knobs2.txt
: > $out
export PM_LIB=$PWD/ocean-perception_amd/lib/libvehicle_pm_gpu_tuning.so
run() { echo "## $1" >> $out; timeout -k 10 300 python tools/stream_matrix.py --legs single,pipe_dev,batch 2>&1 | grep -v amdgpu.ids >> $out; }
run default
for w in 3 5 6; do PM_RUNBLK_WAVES_ROW=$w run "PM_RUNBLK_WAVES_ROW=$w"; PM_RUNBLK_WAVES_COL=$w run "PM_RUNBLK_WAVES_COL=$w"; done
PM_RUNBLK_WAVES_ROW16=5 run "PM_RUNBLK_WAVES_ROW16=5"
PM_RUNBLK_WAVES_COL16=5 run "PM_RUNBLK_WAVES_COL16=5"
PM_G16_ROW_AMP=0.25 run "PM_G16_ROW_AMP=0.25"
PM_G16_COL_AMP=2 run "PM_G16_COL_AMP=2"
PM_G16_ROW_AMP_NEG=4 run "PM_G16_ROW_AMP_NEG=4"
PM_G16_COL_AMP_NEG=8 run "PM_G16_COL_AMP_NEG=8"
run default_again
cat $out
