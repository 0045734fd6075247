#!/bin/bash
# A/B of waves per chain (segments per chain = waves * 64 / group): tools/r03_waves.sh
run() {
  echo -n "$* : "
  env "$@" timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --host-pairs 0 --no-side-legs --no-profile 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f pairs/s  %.3f ms (median %.3f)'%(r['value'], r['ms_per_step'], r['step_ms']['median']))"
}
run PM_RUNBLK_WAVES=4
run PM_RUNBLK_WAVES=6
run PM_RUNBLK_WAVES=8
run PM_RUNBLK_WAVES=12
run PM_RUNBLK_WAVES=16
run PM_RUNBLK_WAVES_ROW=8 PM_RUNBLK_WAVES_COL=4
run PM_RUNBLK_WAVES_ROW=4 PM_RUNBLK_WAVES_COL=8
run PM_RUNBLK_WAVES_ROW=8 PM_RUNBLK_WAVES_ROW16=4 PM_RUNBLK_WAVES_COL=8 PM_RUNBLK_WAVES_COL16=4
run PM_RUNBLK_WAVES_ROW=4 PM_RUNBLK_WAVES_ROW16=8 PM_RUNBLK_WAVES_COL=4 PM_RUNBLK_WAVES_COL16=8
run PM_RUNBLK_GROUP=16
run PM_RUNBLK_GROUP=32
run PM_RUNBLK_GROUP=16 PM_RUNBLK_WAVES=8
run PM_RUNBLK_GROUP=32 PM_RUNBLK_WAVES=8
