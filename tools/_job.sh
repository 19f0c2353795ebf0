cd /root/repo
python3 tools/k1_probe.py > gpurun_out/r05y_k1_new.log 2>&1
tail -1 gpurun_out/r05y_k1_new.log
python -m pytest tests -m gpu -x -q -k "gauss or parity or fixture or golden or ragged or batch" > gpurun_out/r05y_k1_tests.log 2>&1; tail -3 gpurun_out/r05y_k1_tests.log
