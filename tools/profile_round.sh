#!/bin/bash
# Collects the profiles of a round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag>      e.g. r02a  ->  gpurun_out/<tag>_*  (copy what is to be judged into profiles/)
# 1. the bench line (default: three steps in flight, per-kernel figures from un-overlapped steps); 2. rocprofv3 --kernel-trace --stats of
#    the same workload one step at a time (--pipeline 1: what the line's per-kernel figures are measured on); 3./4. separate --pmc passes (FETCH_SIZE, WRITE_SIZE:
# MI355X_MICROARCH.md, HBM section) -> traffic json.
tag=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/${tag}_bench_n1.json 2> $O/${tag}_bench_n1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_stats -o s -- python3 $R/bench.py --steps 3 --warmup 1 --pipeline 1 --no-cpu-baseline > $O/${tag}_stats.log 2>&1
for cnt in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d $O/${tag}_pmc_$cnt -o pmc -- python3 $R/bench.py --steps 1 --warmup 1 --pipeline 1 --no-cpu-baseline > $O/${tag}_pmc_$cnt.log 2>&1
done
python3 $R/tools/pmc_to_traffic.py $O/${tag}_pmc_FETCH_SIZE $O/${tag}_pmc_WRITE_SIZE $O/${tag}_traffic.json $tag > /dev/null
find $O/${tag}_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${tag}_rocprofv3_kernel_stats_bench512.csv
find $O/${tag}_pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $O/${tag}_pmc_FETCH_SIZE_bench512.csv
find $O/${tag}_pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $O/${tag}_pmc_WRITE_SIZE_bench512.csv
tail -c 400 $O/${tag}_bench_n1.json; echo; head -8 $O/${tag}_rocprofv3_kernel_stats_bench512.csv
