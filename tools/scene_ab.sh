#!/bin/bash
# tools/scene_ab.sh "VAR=x" "VAR=y VAR2=z" ...: tools/scene_ab.py once per knob setting ("PM_X=0" = defaults)
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout -k 10 200 python tools/scene_ab.py --steps 12 2>/dev/null
done
