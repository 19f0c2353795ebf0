#!/bin/bash
# round-6 evidence job (one gpurun call): GPU tests, the random campaigns on the final sources (each ends in PASS / FAIL: every image against the glibc build AND the correctly
# rounded restatement, the near-tie property, the margin floor off the allow-list), the profiles of the timed configuration and of one step at a time (tools/profile_round.sh),
# SQ counters, the developer counters' breakdown, the occupancy probe, the sweep over steps in flight and the workspace sizes
#   tools/r06_job_final.sh <tag> [campaign images] [big campaign images]
tag=$1; N=${2:-60000}; NB=${3:-1800}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python -m pytest tests -m gpu -x -q > $O/${tag}_gputests.log 2>&1; echo "gpu tests rc $?"; tail -3 $O/${tag}_gputests.log
python3 tools/campaign.py $N > $O/${tag}_campaign_$N.log 2>&1; echo "campaign rc $?"; tail -3 $O/${tag}_campaign_$N.log | cut -c1-600
CAMPAIGN_BIG=1 python3 tools/campaign.py $NB > $O/${tag}_campaign_big$NB.log 2>&1; echo "big campaign rc $?"; tail -3 $O/${tag}_campaign_big$NB.log | cut -c1-600
python3 tools/campaign_batch.py 60 48 > $O/${tag}_campaign_batch.log 2>&1; tail -2 $O/${tag}_campaign_batch.log
python3 tools/sets_determinism.py > $O/${tag}_sets_determinism.log 2>&1; tail -3 $O/${tag}_sets_determinism.log
tools/profile_round.sh $tag
tools/pmc_valu.sh $tag 1 > $O/${tag}_pmc_valu_768copies.log 2>&1; tail -3 $O/${tag}_pmc_valu_768copies.log
find $O/pmc_valu_${tag} -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $O/${tag}_pmc_SQ_valu_768copies.csv
tools/pmc_batch.sh $tag 4 > $O/${tag}_pmc_batch_w4.log 2>&1; tail -4 $O/${tag}_pmc_batch_w4.log
tools/pmc_k1.sh $tag > $O/${tag}_pmc_frontend_SQ.log 2>&1; tail -6 $O/${tag}_pmc_frontend_SQ.log
LSD_HIP_LIB=$R/linesegmentdetector-slam_amd/liblsdhip_stats.so python3 tools/breakdown.py 4 512 1 > $O/${tag}_breakdown_w4.log 2>&1; tail -36 $O/${tag}_breakdown_w4.log
python3 tools/occupancy_probe.py 8 96 > $O/${tag}_occupancy_probe.log 2>&1; tail -12 $O/${tag}_occupancy_probe.log
for d in 4 6 8 12; do echo "steps in flight $d: $(python3 bench.py --no-cpu-baseline --pipeline $d --steps 96 --warmup 16 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.2f ms per step" % d["ms_per_step"])')"; done > $O/${tag}_depth_sweep.log 2>&1; cat $O/${tag}_depth_sweep.log
python3 tools/workspace_size.py > $O/${tag}_workspace_size.log 2>&1; cat $O/${tag}_workspace_size.log
