cd /root/repo
P=$PWD/linesegmentdetector-slam_amd
export GPU_MAX_HW_QUEUES=8
for lib in exp expA1 expA3 expA5; do
  for feed in 1 2 3 5; do
    echo "== $lib feed $feed"
    LSD_HIP_LIB=$P/liblsdhip_$lib.so LSD_REGION_FEED=$feed timeout 300 python3 tools/breakdown.py 4 512 1 2>&1 | grep "depth 1" | tail -1
    LSD_HIP_LIB=$P/liblsdhip_$lib.so LSD_REGION_FEED=$feed timeout 300 python3 tools/breakdown.py 4 512 8 2>&1 | grep "depth 8" | tail -1
  done
done > gpurun_out/r05y_feed_sweep.log 2>&1
cat gpurun_out/r05y_feed_sweep.log
