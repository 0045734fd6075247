#!/bin/bash
# do the sweep knobs tuned on single pairs (round 3) still hold when two pairs go through every launch together?
mkdir -p gpurun_out/r04
out=gpurun_out/r04/knobs.txt
: > $out
export PM_LIB=$PWD/ocean-perception_amd/lib/libvehicle_pm_gpu_tuning.so
run() { echo "## $1" >> $out; timeout -k 10 300 python tools/stream_matrix.py --legs single,pipe_dev,batch 2>&1 | grep -v amdgpu.ids >> $out; }
run default
for w in 3 5 6 8; do PM_RUNBLK_WAVES=$w run "PM_RUNBLK_WAVES=$w"; done
for g in 16 32; do PM_RUNBLK_GROUP=$g run "PM_RUNBLK_GROUP=$g"; done
PM_PAIR_CHUNK=3 run "PM_PAIR_CHUNK=3"
PM_PAIR_CHUNK=1 run "PM_PAIR_CHUNK=1"
PM_G16_ROW_AMP=2 PM_G16_COL_AMP=8 run "G16 thresholds one octave up (forward)"
PM_G16_ROW_AMP_NEG=16 PM_G16_COL_AMP_NEG=32 run "G16 thresholds one octave up (backward)"
PM_RUN2_LREF=0 run "PM_RUN2_LREF=0"
PM_RUN2_LREF=3 PM_RUN2_LREF_KB=64 run "PM_RUN2_LREF=3 (rows too)"
cat $out
