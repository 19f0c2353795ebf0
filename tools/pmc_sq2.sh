#!/bin/bash
# developer probe: instruction-cache / branch / issue counters of the region stage on one heavy bench image (run on the GPU box)
#   tools/pmc_sq2.sh <tag> [waves] [image ids...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"; do
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
