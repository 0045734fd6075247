#!/bin/bash
# round 5, call c: new tests, the reference-shape leg, plane mode with / without early termination, default bench
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/r05
mkdir -p $out
cd $root
python -m pytest tests/test_streams.py tests/test_dist.py tests/test_gpu_parity.py -m gpu -x -q -k "streams or stream_priority or small or producers_event or held_frame or sequence or rccl" > $out/c_tests.log 2>&1
tail -4 $out/c_tests.log
python tools/r05_refshape.py 2>/dev/null > $out/c_refshape.json
cat $out/c_refshape.json
for lib in libvehicle_pm_gpu.so libvehicle_pm_gpu_tuning.so; do
  for m in "" "--state f16 --enhance"; do
    PM_LIB=$root/ocean-perception_amd/lib/$lib python bench.py --mode planes $m --steps 12 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$m', 'pairs/s', round(j['value'],1), 'ms', round(j['ms_per_step'],3), j['check'])"
  done
done
python bench.py > $out/c_bench.json 2> $out/c_bench.err
python -c "
import json
j=json.load(open('$out/c_bench.json'))
print('headline', j['value'], j['check'].get('equals_oracle_full_frame'))
print('seqdev', j['sequence_device']['value'], 'hostseq', j['host_sequence_all_ranks']['value'])
print('ref', {k:v for k,v in j['reference_test_shape'].items() if 'ms_per' in k or k=='views_on_two_streams'})
"
