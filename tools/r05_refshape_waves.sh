#!/bin/bash
# the reference's own 376x240 call with more wavefronts per chain (tuning build): the chip is nearly empty at this size
root=$GRAFT_REPO_ROOT
cd $root
T=$root/ocean-perception_amd/lib/libvehicle_pm_gpu_tuning.so
for w in 0 6 8 12 16; do
  PM_LIB=$T PM_RUNBLK_WAVES=$w python tools/r05_refshape.py 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('PM_RUNBLK_WAVES=$w', j['ms_per_call_first_five'], round(j['ms_per_call_steady_median'],4), j['equals_the_golden_row_checksums'])"
done
for g in 8 32; do
  PM_LIB=$T PM_RUNBLK_GROUP=$g python tools/r05_refshape.py 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('PM_RUNBLK_GROUP=$g', round(j['ms_per_call_steady_median'],4), j['equals_the_golden_row_checksums'])"
done
