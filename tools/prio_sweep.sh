#!/bin/bash
# developer probe: ms per step of a long timed region against steps in flight, hardware queues and the front-end priority experiment
#   tools/prio_sweep.sh        (run through gpurun from the repo root; writes gpurun_out/prio_sweep.log)
R=$GRAFT_REPO_ROOT
run() {  # queues depth front-priority
  r=$(GPU_MAX_HW_QUEUES=$1 LSD_FRONT_PRIORITY=$3 timeout 300 python3 $R/bench.py --no-cpu-baseline --pipeline $2 --steps 96 --warmup $((2 * $2)) 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f' % j['ms_per_step'])")
  echo "queues $1 depth $2 front-priority $3: $r ms/step"
}
{
run 8 8 0
run 8 8 3
run 16 8 3
run 16 8 4
run 8 8 4
run 24 12 3
run 8 8 0
} > $R/gpurun_out/prio_sweep.log 2>&1
cat $R/gpurun_out/prio_sweep.log
