#!/bin/bash
# Collects the profiles of a round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag>      e.g. r04a  ->  gpurun_out/<tag>_*  (copy what is to be judged into profiles/)
# 1. rocprofv3 --kernel-trace --stats of the TIMED configuration (bench.py's defaults, --no-cpu-baseline)  -> <tag>_w4_pipeline8_*
#    and of the same workload one step at a time (--pipeline 1: library defaults, 8 waves, help on)         -> <tag>_w8_pipeline1_*
# 2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE: MI355X_MICROARCH.md, HBM section) of both -> traffic json
#    (rocprofv3 serialises the dispatches of a --pmc pass, so the per-launch traffic of the overlapped configuration is that of
#     its kernels run one after the other).
# 3. last, the bench line (`python3 bench.py`, defaults: 20 timed steps, eight steps in flight on the 4-wave region stage, help off;
#    per-kernel figures of the line from un-overlapped steps right after the timed region) and the same with 128 timed steps: the
#    lines quote the PMC traffic of THESE sources (profiles/traffic_latest.json carries the sources' sha), so the passes come first.
tag=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for cfg in w4_pipeline8 w8_pipeline1; do
  if [ $cfg = w4_pipeline8 ]; then args="--steps 32 --warmup 16"; else args="--steps 3 --warmup 1 --pipeline 1"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_${cfg}_stats -o s -- python3 $R/bench.py $args --no-cpu-baseline > $O/${tag}_${cfg}_stats.log 2>&1
  find $O/${tag}_${cfg}_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${tag}_${cfg}_rocprofv3_kernel_stats_bench512.csv
  if [ $cfg = w4_pipeline8 ]; then args="--steps 8 --warmup 8"; else args="--steps 1 --warmup 1 --pipeline 1"; fi
  for cnt in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d $O/${tag}_${cfg}_pmc_$cnt -o pmc -- python3 $R/bench.py $args --no-cpu-baseline > $O/${tag}_${cfg}_pmc_$cnt.log 2>&1
    find $O/${tag}_${cfg}_pmc_$cnt -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $O/${tag}_${cfg}_pmc_${cnt}_bench512.csv
  done
  python3 $R/tools/pmc_to_traffic.py $O/${tag}_${cfg}_pmc_FETCH_SIZE $O/${tag}_${cfg}_pmc_WRITE_SIZE $O/${tag}_${cfg}_traffic.json ${tag}_${cfg} > /dev/null
done
cp $O/${tag}_w8_pipeline1_traffic.json $R/profiles/traffic_latest.json
python3 $R/bench.py > $O/${tag}_bench_n1.json 2> $O/${tag}_bench_n1.err                                      # 20 timed steps, 5 warm-up: the driver's invocation
python3 $R/bench.py --steps 128 --warmup 16 --no-cpu-baseline > $O/${tag}_bench_n1_steps128.json 2> /dev/null   # the long-run rate (the drain amortised)
tail -c 600 $O/${tag}_bench_n1.json; echo
head -8 $O/${tag}_w4_pipeline8_rocprofv3_kernel_stats_bench512.csv; head -8 $O/${tag}_w8_pipeline1_rocprofv3_kernel_stats_bench512.csv
