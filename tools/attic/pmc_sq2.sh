#!/bin/bash
# developer probe: instruction-cache / branch / issue counters of the region stage on one heavy bench image (run on the GPU box)
#   tools/pmc_sq2.sh <tag> [waves] [image ids...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  n=$(echo $set | cut -d' ' -f1)
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq2_${tag}_$n
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out -o pmc -- python3 $GRAFT_REPO_ROOT/tools/one_stats.py "$@" > $out.log 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$out/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for p in f:
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "region" in k:
        print(k, {c: round(v / cnt[(k, c)]) for c, v in d.items()})
PY
done
