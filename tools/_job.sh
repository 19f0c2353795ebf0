cd /root/repo
P=$PWD/linesegmentdetector-slam_amd
export GPU_MAX_HW_QUEUES=8
for big in 2 3 4 6; do
  echo "== BIG $big"
  LSD_HIP_LIB=$P/liblsdhip_exp.so LSD_REGION_BIG=$big timeout 300 python3 tools/breakdown.py 4 512 1 2>&1 | grep "depth 1" | tail -1
  LSD_HIP_LIB=$P/liblsdhip_exp.so LSD_REGION_BIG=$big timeout 300 python3 tools/breakdown.py 4 512 8 2>&1 | grep "depth 8" | tail -1
  LSD_HIP_LIB=$P/liblsdhip_exp.so LSD_REGION_BIG=$big timeout 300 python3 tools/breakdown.py 8 512 1 2>&1 | grep "depth 1" | tail -1
done
for soft in 96 384; do for claim in 384 1536; do
  echo "== SOFT $soft CLAIM $claim"
  LSD_HIP_LIB=$P/liblsdhip_exp.so LSD_REGION_SOFT=$soft LSD_REGION_CLAIM=$claim timeout 300 python3 tools/breakdown.py 4 512 8 2>&1 | grep "depth 8" | tail -1
done; done
