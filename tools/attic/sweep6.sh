for cfg in "32 96" "64 96" "32 192" "16 96" "64 32" "32 32"; do
  set -- $cfg
  echo "== up $1 down $2"
  LSD_REGION_UP=$1 LSD_REGION_DOWN=$2 LSD_REGION_HELP=0 HELPSTATS=1 timeout 120 python tools/hang_probe.py 512 2048 2 2>&1 | grep -v amdgpu.ids | grep "rep\|help:"
  LSD_REGION_UP=$1 LSD_REGION_DOWN=$2 LSD_REGION_HELP=0 timeout 120 python tools/one_stats.py 8 0 1 110 27 45 77 187 2>&1 | grep -v amdgpu.ids | awk '{printf " %s:%s", $1, $6} END {print ""}'
done
