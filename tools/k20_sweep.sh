#!/bin/bash
# developer probe: the driver's invocation (--steps 20 --warmup 5) against the number of hardware queues, three runs each
R=$GRAFT_REPO_ROOT
for q in 8 32 16 8 32; do for rep in 1 2 3; do
  r=$(GPU_MAX_HW_QUEUES=$q python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms/step (%.1f Gpix/s), one step %.1f' % (j['ms_per_step'], j['value']/1e3, j['one_step_at_a_time']['ms_per_step']))")
  echo "queues $q: $r"
done; done
