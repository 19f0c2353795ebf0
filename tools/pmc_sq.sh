#!/bin/bash
# developer probe: SQ instruction-mix counters of the region stage on one heavy bench image (run on the GPU box)
#   tools/pmc_sq.sh <tag> [image ids...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq_$tag
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  --kernel-trace --output-format csv -d $out -o pmc -- python3 $GRAFT_REPO_ROOT/tools/one_stats.py "$@" > $out.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$out/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for p in f:
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "region" in k:
        n = cnt[(k, "SQ_INSTS_VALU")]
        print(k, "dispatches", n, {c: round(v / n) for c, v in d.items()})
PY
tail -5 $out.log
