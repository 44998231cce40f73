#!/bin/bash
# A/B of one environment switch at configs[1] (8x4, m=512): tools/ab_env_m512.sh VAR A B ...   (prints sweep phase means per value)
var=$1; shift
for v in "$@"; do
  echo "== $var=$v"
  env $var=$v timeout -k 10 300 python3 tools/sweep_timing.py -Lx 8 -Ly 4 -J1 1 -Jz1 1 -J2 0.5 -Jz2 0.5 -mwarmup 512 -nsweeps 2 2>&1 | grep "sweep steps" || exit 1
done
