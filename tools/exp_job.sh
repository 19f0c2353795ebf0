#!/bin/bash
# experiment job (GPU box): parity subset + bench of an experiment build of the region stage against the product build
#   tools/exp_job.sh <tag> [full]     (the experiment library: linesegmentdetector-slam_amd/liblsdhip_exp.so)
tag=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; EXP=$R/linesegmentdetector-slam_amd/liblsdhip_exp.so
cd $R
if [ "$2" = full ]; then python -m pytest tests -m gpu -x -q > $O/${tag}_gputests.log 2>&1; echo "product gpu tests rc $?"; tail -3 $O/${tag}_gputests.log; fi
LSD_HIP_LIB=$EXP python -m pytest tests -m gpu -x -q -k "fixture_parity or bench_batch_sample or giant or variants or stamp or timed_configuration or synthetic_ragged or other_parameters or schedule" > $O/${tag}_exp_tests.log 2>&1; echo "exp tests rc $?"; tail -3 $O/${tag}_exp_tests.log
python3 bench.py --no-cpu-baseline > $O/${tag}_bench_prod.json 2> $O/${tag}_bench_prod.err
LSD_HIP_LIB=$EXP python3 bench.py --no-cpu-baseline > $O/${tag}_bench_exp.json 2> $O/${tag}_bench_exp.err
python3 - <<PY
import json
for k in ("prod", "exp"):
    try:
        j = json.load(open("$O/${tag}_bench_%s.json" % k))
        print(k, "ms/step %.2f  one-at-a-time %.2f  kernels %s  timed cyc mean %.1fM max %.1fM" % (j["ms_per_step"], j["one_step_at_a_time"]["ms_per_step"],
              {a: round(b, 2) for a, b in j["kernel_ms"].items()}, j["dominant_kernel"]["timed_region"]["cycles_per_image"]["mean"] / 1e6, j["dominant_kernel"]["timed_region"]["cycles_per_image"]["max"] / 1e6))
    except Exception as e:
        print(k, "failed", e); print(open("$O/${tag}_bench_%s.err" % k).read()[-1500:])
PY
for lib in "" $EXP; do LSD_HIP_LIB=$lib python3 tools/one_stats.py 4 0 1 110 45 187 2>&1 | grep "waves" | cut -c1-60; done
