cd /root/repo
python -m pytest tests -m gpu -x -q > gpurun_out/r05y_nfa_tests.log 2>&1; tail -3 gpurun_out/r05y_nfa_tests.log
LSD_HIP_LIB=$PWD/linesegmentdetector-slam_amd/liblsdhip_stats.so python3 tools/breakdown.py 4 > gpurun_out/r05y_breakdown_nfa2.log 2>&1
grep -n "per-image\|nfa\|cycles_eval \|cycles_grow" gpurun_out/r05y_breakdown_nfa2.log
