#!/bin/bash
# developer probe: ms per step of the timed region against steps in flight and hardware queues    tools/depth_sweep.sh [lib]
R=$GRAFT_REPO_ROOT
for q in 4 8 16; do for d in 4 6 8; do
  r=$(GPU_MAX_HW_QUEUES=$q LSD_HIP_LIB=$1 python3 $R/bench.py --no-cpu-baseline --pipeline $d --steps 24 --warmup 8 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f' % j['ms_per_step'])")
  echo "queues $q depth $d: $r ms/step"
done; done
