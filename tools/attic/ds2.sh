R=$GRAFT_REPO_ROOT
for cfg in "12 10" "12 12" "16 16"; do set -- $cfg
  r=$(GPU_MAX_HW_QUEUES=$1 LSD_HIP_LIB=$R/linesegmentdetector-slam_amd/liblsdhip_exp.so python3 $R/bench.py --no-cpu-baseline --pipeline $2 --steps 32 --warmup 16 2>/tmp/err.log | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f' % j['ms_per_step'])")
  echo "queues $1 depth $2: $r ms/step"
tail -3 /tmp/err.log; done
