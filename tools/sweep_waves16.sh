#!/bin/bash
# waves per chain by group width (GPU box): row32 row16 col32 col16
run() {
  python3 bench.py --no-side-legs --no-profile --steps 30 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  %.1f pairs/s  %.3f ms' % (r['value'], r['ms_per_step']))"
}
for cfg in "4 4 4 4" "4 2 4 4" "4 3 4 4" "4 4 4 2" "4 4 4 3" "4 3 4 3" "4 2 4 2" "3 4 4 4" "4 4 3 4" "5 4 4 4" "4 5 4 5"; do
  set -- $cfg
  echo "waves row32 $1 row16 $2 col32 $3 col16 $4"; PM_RUNBLK_WAVES_ROW=$1 PM_RUNBLK_WAVES_ROW16=$2 PM_RUNBLK_WAVES_COL=$3 PM_RUNBLK_WAVES_COL16=$4 run
done
