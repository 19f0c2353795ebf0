#!/bin/bash
# developer probe: the product library against the experiment build (make exp) on the timed configuration
R=$GRAFT_REPO_ROOT
run() {  # label lib queues depth
  r=$(GPU_MAX_HW_QUEUES=$3 LSD_HIP_LIB=$2 timeout 300 python3 $R/bench.py --no-cpu-baseline --pipeline $4 --steps 96 --warmup $((2 * $4)) 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms/step; one step at a time %.1f; kernels %s' % (j['ms_per_step'], j.get('one_step_ms', j.get('unoverlapped_ms_per_step', 0)), j.get('kernel_ms')))")
  echo "$1 queues $3 depth $4: $r"
}
{
run product "" 8 8
run exp $R/linesegmentdetector-slam_amd/liblsdhip_exp.so 8 8
run exp $R/linesegmentdetector-slam_amd/liblsdhip_exp.so 12 12
run product "" 8 8
run exp $R/linesegmentdetector-slam_amd/liblsdhip_exp.so 8 8
} > $R/gpurun_out/exp_job.log 2>&1
cat $R/gpurun_out/exp_job.log
