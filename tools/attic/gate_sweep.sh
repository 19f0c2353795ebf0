for g in 0 12000 35000 70000; do echo "== gate $g"; LSD_REGION_GATE=$g python3 tools/single_step_probe.py 2>&1 | grep "waves\|images" | grep -v "help  0\|help   0\|help  64"; done
