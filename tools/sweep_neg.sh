#!/bin/bash
# 16-lane-group amplitude thresholds, backward sweeps separately (GPU box): row_fwd col_fwd row_neg col_neg
run() {
  python3 bench.py --no-side-legs --no-profile --steps 30 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  %.1f pairs/s  %.3f ms' % (r['value'], r['ms_per_step']))"
}
for cfg in "0.5 4 4 16" "0.5 4 8 32" "0.5 4 16 16" "0.5 4 16 32" "0.5 4 8 64" "0.5 4 4 64" "1 4 8 32" "0.5 8 8 32" "0.25 4 8 32" "0.5 2 8 32"; do
  set -- $cfg
  echo "row $1 col $2 row_neg $3 col_neg $4"; PM_G16_ROW_AMP=$1 PM_G16_COL_AMP=$2 PM_G16_ROW_AMP_NEG=$3 PM_G16_COL_AMP_NEG=$4 run
done
