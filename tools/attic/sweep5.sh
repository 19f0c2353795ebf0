for cfg in "384 1536 0" "384 1024 0" "512 1536 0" "384 1536 24"; do
  set -- $cfg
  echo "== soft $1 claim $2 help $3"
  LSD_REGION_SOFT=$1 LSD_REGION_CLAIM=$2 LSD_REGION_HELP=$3 HELPSTATS=1 timeout 120 python tools/hang_probe.py 512 2048 1 2>&1 | grep -v amdgpu.ids
  LSD_REGION_HELP=0 LSD_REGION_SOFT=$1 LSD_REGION_CLAIM=$2 timeout 120 python tools/one_stats.py 8 0 1 110 27 45 77 187 2>&1 | grep -v amdgpu.ids | awk '{print "   img", $1, $4, $5, $6}'
done
