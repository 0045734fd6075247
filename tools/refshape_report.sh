#!/bin/bash
# Everything profiles/r06_reference_call_*.txt is made of, on one box: tools/refshape_report.sh
# (needs the analysis builds: tools/build_variant.sh phases1 pm_sweeps -DPM_RUN2_PHASES=1; phases2 ... =2; hostphases pm_hostpath -DPM_HOST_PHASES)
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/${ROUND:-r06}
mkdir -p $out
cd $root
L=ocean-perception_amd/lib
{
  echo "## steady-state call, product library (median of 300 calls, three processes)"
  for i in 1 2 3; do CALLS=300 python tools/ref_shape_loop.py 2>&1 | grep median; done
  echo "## run-step counters of one Match (pm_debug_counters)"
  python tools/refshape_steps.py 2>&1 | grep sweeps
  echo "## phases of a chain's workgroup, device wall clock (analysis builds -DPM_RUN2_PHASES)"
  PHASES=1 PM_LIB=$L/libvehicle_pm_gpu_phases1.so python tools/refshape_steps.py 2>&1 | grep sweeps
  PHASES=2 PM_LIB=$L/libvehicle_pm_gpu_phases2.so python tools/refshape_steps.py 2>&1 | grep sweeps
  echo "## where the HOST spends a call (analysis build -DPM_HOST_PHASES; the last line is steady state)"
  CALLS=80 PM_LIB=$L/libvehicle_pm_gpu_hostphases.so python tools/ref_shape_loop.py 2>&1 | grep "host phases" | tail -1
} > $out/refshape_report.txt 2>&1
python tools/run_structure.py --out $out/run_structure_ref.txt > /dev/null 2>&1
bash tools/ref_shape_trace.sh > /dev/null 2>&1
cat $out/refshape_report.txt
