#!/bin/bash
# the round's closing job on the final sources (after tools/r06_job_final.sh: K1's tile height and bench.py's extras changed since): GPU tests, a short pair of campaigns,
# the profiles of the timed configuration and of one step at a time with their PMC passes, the bench lines
#   tools/r06_job_short.sh <tag>
tag=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python -m pytest tests -m gpu -x -q > $O/${tag}_gputests.log 2>&1; echo "gpu tests rc $?"; tail -3 $O/${tag}_gputests.log
python3 tools/campaign.py 6000 > $O/${tag}_campaign_6000.log 2>&1; echo "campaign rc $?"; tail -1 $O/${tag}_campaign_6000.log | cut -c1-300
CAMPAIGN_BIG=1 python3 tools/campaign.py 200 > $O/${tag}_campaign_big200.log 2>&1; echo "big campaign rc $?"; tail -1 $O/${tag}_campaign_big200.log | cut -c1-300
tools/profile_round.sh $tag
tools/pmc_k1.sh $tag > $O/${tag}_pmc_frontend_SQ.log 2>&1; tail -6 $O/${tag}_pmc_frontend_SQ.log | cut -c1-400
