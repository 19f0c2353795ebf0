for cfg in "384 384 3" "1536 1536 12" "1024 1024 8" "768 768 6" "1900 1900 16"; do
  set -- $cfg
  echo "== soft $1 claim $2 big $3"
  LSD_REGION_HELP=0 LSD_REGION_SOFT=$1 LSD_REGION_CLAIM=$2 LSD_REGION_BIG=$3 timeout 120 python tools/hang_probe.py 512 2048 2 2>&1 | grep rep
  LSD_REGION_HELP=0 LSD_REGION_SOFT=$1 LSD_REGION_CLAIM=$2 LSD_REGION_BIG=$3 timeout 120 python tools/one_stats.py 8 0 1 110 27 187 2>&1 | grep -v amdgpu.ids | awk '{print "   img", $1, $4, $5, $6}'
done
