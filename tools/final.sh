#!/bin/bash
# Final measurements of a round on the GPU box (one call): ROUND=r06 tools/final.sh <head-sha> [steps...]
#   tests    the whole -m gpu suite
#   fuzz     the six differential fuzzers, new seeds            -> gpurun_out/$ROUND/fuzz.txt
#   bench    the driver's command                                -> gpurun_out/$ROUND/bench_default.json
#   prof     tools/profile.sh                                    -> gpurun_out/prof_$ROUND/
head=${1:-unknown}; shift
steps=${@:-tests fuzz bench prof}
root=$GRAFT_REPO_ROOT
export ROUND=${ROUND:-r06}
out=$root/gpurun_out/$ROUND
mkdir -p $out
cd $root
for s in $steps; do
case $s in
tests)
  python -m pytest tests -m gpu -x -q > $out/final_tests.log 2>&1; rc=$?; tail -3 $out/final_tests.log; [ $rc -eq 0 ] || exit 1;;
fuzz)
  f=$out/fuzz.txt
  echo "Differential fuzzers on the tree of commit $head (tools/final.sh fuzz, one MI355X; every case bit-identical or the run stops)" > $f
  run() { name=$1; shift; echo "## $name $*" >> $f; timeout -k 10 900 python tools/$name "$@" > $out/fuzz_$name.log 2>&1; rc=$?; tail -1 $out/fuzz_$name.log >> $f; echo "exit $rc" >> $f; echo "$name done ($rc)"; [ $rc -eq 0 ]; }
  run fuzz_api.py --cases 500 --seed 661 || { cat $f; exit 1; }
  run fuzz_engines.py --cases 400 --seed 662 || { cat $f; exit 1; }
  run fuzz_engines.py --cases 16 --seed 667 --big || { cat $f; exit 1; }
  run fuzz_planes.py --cases 500 --seed 663 || { cat $f; exit 1; }
  run fuzz_seed.py --cases 300 --seed 664 || { cat $f; exit 1; }
  run fuzz_selfseed.py --cases 200 --seed 665 || { cat $f; exit 1; }
  run fuzz_tiled.py --cases 250 --seed 666 || { cat $f; exit 1; }
  cat $f;;
bench)
  python bench.py > $out/bench_default.json 2> $out/bench_default.err
  python -c "
import json
j=json.load(open('$out/bench_default.json'))
print('headline', round(j['value'],1), j['check'].get('equals_oracle_full_frame'))
print('roofline frac', j['roofline']['frac'], 'traffic', j['roofline']['traffic'], 'alg', j['roofline']['algorithmic_bytes_per_launch'])
print('batch', {k:(round(v['value'],1), v['check']['passes']) for k,v in j['batch'].items()})
print('seqdev', round(j['sequence_device']['value'],1), 'hostseq', round(j['host_sequence_all_ranks']['value'],1))
print('planes', {k:(round(v['value'],1), v['check'].get('equals_oracle_full_frame',{}).get('differing_pixels')) for k,v in j['planes'].items() if 'value' in v and 'check' in v})
print('ref', j['reference_test_shape']['ms_per_call_steady_median'], 'tiled', j['tiled_4096x2160']['ms_per_frame'], j['tiled_4096x2160']['eight_bands_on_this_device']['ms_per_frame'])
";;
prof)
  bash tools/profile.sh $head > $out/profile.log 2>&1; tail -20 $out/profile.log;;
esac
done
