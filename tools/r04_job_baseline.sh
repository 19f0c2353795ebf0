#!/bin/bash
# round-4 job: GPU tests + random campaigns + the profiles of the timed configuration, one gpurun call
#   tools/r04_job_baseline.sh <tag> [campaign images] [big campaign images]
tag=$1; N=${2:-20000}; NB=${3:-600}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python -m pytest tests -m gpu -x -q > $O/${tag}_gputests.log 2>&1; echo "gpu tests rc $?"; tail -3 $O/${tag}_gputests.log
[ $N -gt 0 ] && { python3 tools/campaign.py $N > $O/${tag}_campaign_$N.log 2>&1; tail -2 $O/${tag}_campaign_$N.log; }
[ $NB -gt 0 ] && { CAMPAIGN_BIG=1 python3 tools/campaign.py $NB > $O/${tag}_campaign_big$NB.log 2>&1; tail -2 $O/${tag}_campaign_big$NB.log; }
python3 tools/campaign_batch.py 60 48 > $O/${tag}_campaign_batch.log 2>&1; tail -2 $O/${tag}_campaign_batch.log
tools/profile_round.sh $tag
tools/pmc_sq.sh ${tag}_w8_187 8 187
tools/pmc_sq.sh ${tag}_w4_187 4 187
for v in w8 w4; do find $O/pmc_sq_${tag}_${v}_187 -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $O/${tag}_pmc_SQ_insts_image187_${v}.csv; done
