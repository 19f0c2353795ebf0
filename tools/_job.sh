cd /root/repo
P=$PWD/linesegmentdetector-slam_amd
export GPU_MAX_HW_QUEUES=8
for lib in exp4rw512 exp3; do
  echo "== $lib"
  for i in 1 2; do
  LSD_HIP_LIB=$P/liblsdhip_$lib.so timeout 300 python3 tools/breakdown.py 4 512 1 2>&1 | grep "depth 1\|per-image" | tail -2
  LSD_HIP_LIB=$P/liblsdhip_$lib.so timeout 300 python3 tools/breakdown.py 4 512 8 2>&1 | grep "depth 8" | tail -1
  done
done
