#!/bin/bash
# developer probe: help across workgroups in throughput mode for the few images that run long (gate in ms of running time before an image asks)
R=$GRAFT_REPO_ROOT
# LSD_REGION_<NAME> is read by the DEVELOPER builds only (csrc/Makefile: make stats); the shipped library ignores it with a note on stderr
export LSD_HIP_LIB=${LSD_HIP_LIB:-$GRAFT_REPO_ROOT/linesegmentdetector-slam_amd/liblsdhip_stats.so}
[ -f "$LSD_HIP_LIB" ] || make -C $GRAFT_REPO_ROOT/linesegmentdetector-slam_amd/csrc stats >/dev/null
run() {  # gate_ms help_waves steps
  r=$(LSD_REGION_LINGER=${LINGER:-1000000} LSD_REGION_GATE=$(( $1 * 2344 )) python3 $R/bench.py --no-cpu-baseline --help-waves $2 --steps $3 --warmup $4 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=j['dominant_kernel']['timed_region']['cycles_per_image']; print('%.2f ms/step; last timed step: Mcycles per image mean %.0f max %.0f' % (j['ms_per_step'], t['mean']/1e6, t['max']/1e6))")
  echo "linger $LINGER gate $1 ms help $2 steps $3: $r"
}
export LINGER=${LINGER:-1}
run 0 0 96 16
for g in 30 60 90; do for h in 8 24; do run $g $h 96 16; done; done
run 0 0 20 5
for g in 30 60 90; do for h in 8 24; do run $g $h 20 5; done; done
