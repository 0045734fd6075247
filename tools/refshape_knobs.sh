#!/bin/bash
# The reference's own call (tools/ref_shape_loop.py) under group-width / wavefront-count knobs of the PM_SEM_GPU sweeps,
# forward and backward sweeps separately (tuning build).
lib=ocean-perception_amd/lib/libvehicle_pm_gpu_tuning.so
run() { echo -n "$* : "; env "$@" PM_LIB=$lib python tools/ref_shape_loop.py 2>&1 | grep median; }
run X=0
for gf in 16 32; do for gb in 8 16 32; do for wf in 2 4 8; do for wb in 2 4 8; do
  run PM_GPU_GROUP_FWD=$gf PM_GPU_GROUP_BWD=$gb PM_GPU_WAVES_FWD=$wf PM_GPU_WAVES_BWD=$wb
done; done; done; done
