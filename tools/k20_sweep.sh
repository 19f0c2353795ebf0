#!/bin/bash
# developer probe: the driver's invocation (--steps 20 --warmup 5) under the tail settings
R=$GRAFT_REPO_ROOT
for cfg in "8 0" "8 1" "8 2" "8 3" "7 0" "10 0" "8 0"; do set -- $cfg
  r=$(python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --pipeline $1 --tail-help $2 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms/step (%.1f Gpix/s), one step %.1f' % (j['ms_per_step'], j['value']/1e3, j['one_step_at_a_time']['ms_per_step']))")
  echo "depth $1 tail-help $2: $r"
done
