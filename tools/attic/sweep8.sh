for cfg in "10 20000" "5 20000" "2 20000" "10 100000" "5 100000" "10 400000"; do
  set -- $cfg
  echo "== wb $1 xpoll $2"
  for img in 45 187; do LSD_REGION_WB=$1 LSD_REGION_XPOLL=$2 timeout 120 python tools/help_probe.py $img 0 256 2>&1 | grep region | tail -2 | cut -c1-110; done
  LSD_REGION_WB=$1 LSD_REGION_XPOLL=$2 HELPSTATS=1 timeout 120 python tools/hang_probe.py 512 2048 3 2>&1 | grep "rep\|help:"
done
