#!/bin/bash
# Developer probe: region-stage time of the bench batch under different schedule parameters (LSD_REGION_* environment variables).
for cfg in "512 384 3 0" "512 512 3 0" "1024 1024 3 0" "256 256 3 0" "512 384 1 0" "512 384 5 0" "512 384 3 1" "2048 2048 3 0" "384 192 3 0"; do
  set -- $cfg
  echo "== soft $1 claim $2 feed $3 big $4"
  LSD_REGION_SOFT=$1 LSD_REGION_CLAIM=$2 LSD_REGION_FEED=$3 LSD_REGION_BIG=$4 timeout 120 python tools/hang_probe.py 512 2048 3 2>&1 | grep -v amdgpu.ids | awk '{print "   ", $0}'
  LSD_REGION_SOFT=$1 LSD_REGION_CLAIM=$2 LSD_REGION_FEED=$3 LSD_REGION_BIG=$4 timeout 120 python tools/one_stats.py 8 0 1 16 187 2>&1 | grep -v amdgpu.ids | awk '{print "   img", $1, $4, $5, $6}'
done
