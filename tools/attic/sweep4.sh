for cfg in "384 3 24 0" "384 3 24 8" "384 3 24 16" "1024 8 24 0" "1024 8 24 16" "384 3 48 32"; do
  set -- $cfg
  echo "== soft $1 big $2 help $3 early $4"
  LSD_REGION_EARLY=$4 LSD_REGION_HELP=$3 LSD_REGION_SOFT=$1 LSD_REGION_CLAIM=$1 LSD_REGION_BIG=$2 HELPSTATS=1 timeout 120 python tools/hang_probe.py 512 2048 2 2>&1 | grep -v amdgpu.ids | grep -v "      image"
done
