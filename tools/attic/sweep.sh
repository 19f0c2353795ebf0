#!/bin/bash
# Developer probe: region-stage time of the bench batch under different schedule parameters (LSD_REGION_* environment variables:
# SOFT = initial / minimal look-ahead, CLAIM = maximal look-ahead, FEED = idle groups per refill).
for cfg in "$@"; do
  set -- $cfg
  echo "== soft $1 claim $2 feed $3"
  LSD_REGION_SOFT=$1 LSD_REGION_CLAIM=$2 LSD_REGION_FEED=$3 timeout 120 python tools/hang_probe.py 512 2048 3 2>&1 | grep -v amdgpu.ids | awk '{print "   ", $0}'
  LSD_REGION_SOFT=$1 LSD_REGION_CLAIM=$2 LSD_REGION_FEED=$3 timeout 120 python tools/one_stats.py 8 0 1 16 187 2>&1 | grep -v amdgpu.ids | awk '{print "   img", $1, $4, $5, $6}'
done
