#!/bin/bash
# developer probe: one 512-image batch alone (8 waves, help on) with EARLY helpers (finished workgroups help although others still wait
# for a CU) and the running-time gate an image must pass before it asks
# LSD_REGION_<NAME> is read by the DEVELOPER builds only (csrc/Makefile: make stats); the shipped library ignores it with a note on stderr
export LSD_HIP_LIB=${LSD_HIP_LIB:-$GRAFT_REPO_ROOT/linesegmentdetector-slam_amd/liblsdhip_stats.so}
[ -f "$LSD_HIP_LIB" ] || make -C $GRAFT_REPO_ROOT/linesegmentdetector-slam_amd/csrc stats >/dev/null
for cfg in "0 12000 0" "0 12000 75" "16 12000 75" "32 12000 75" "32 12000 85" "48 12000 75" "64 12000 80"; do set -- $cfg
  echo "early $1 gate $2 share $3: $(LSD_REGION_EARLY=$1 LSD_REGION_GATE=$2 LSD_REGION_SHARE=$3 python3 $GRAFT_REPO_ROOT/tools/single_step_probe.py 2>&1 | grep 'waves 8 help  -1' | cut -c1-150)"
done
