#!/bin/bash
# waves-per-chain (row / column sweeps) and 16-lane-group amplitude thresholds at the default bench (GPU box)
run() {
  python3 bench.py --no-side-legs --no-profile --steps 30 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  %.1f pairs/s  %.3f ms' % (r['value'], r['ms_per_step']))"
}
for wr in 4 5 6 8; do
  for wc in 3 4 6; do
    echo "waves row $wr col $wc"; PM_RUNBLK_WAVES_ROW=$wr PM_RUNBLK_WAVES_COL=$wc run
  done
done
for ar in 0.5 2 8; do
  for ac in 1 4 16; do
    echo "g16 amp row $ar col $ac"; PM_G16_ROW_AMP=$ar PM_G16_COL_AMP=$ac run
  done
done
