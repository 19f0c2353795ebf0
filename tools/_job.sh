cd /root/repo
P=$PWD/linesegmentdetector-slam_amd
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05g_scan_tests.log 2>&1; tail -3 gpurun_out/r05g_scan_tests.log
export GPU_MAX_HW_QUEUES=8
for i in 1 2; do
timeout 300 python3 tools/breakdown.py 4 512 1 2>&1 | grep "depth 1\|per-image" | tail -2
timeout 300 python3 tools/breakdown.py 4 512 8 2>&1 | grep "depth 8" | tail -1
done
python3 tools/one_stats.py 8 27 100 2>&1 | grep waves | cut -c1-60
python3 tools/one_stats.py 4 27 100 2>&1 | grep waves | cut -c1-60
