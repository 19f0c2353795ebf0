#!/bin/bash
# developer probe: SQ counters of the front end's kernels on the bench batch (run on the GPU box)   tools/pmc_k1.sh <tag>
tag=$1
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_INSTS_BRANCH"; do
  i=$((i+1))
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_k1_${tag}_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out -o pmc -- python3 $GRAFT_REPO_ROOT/tools/k1_probe.py > $out.log 2>&1
  tail -2 $out.log
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for p in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc_k1_${tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0][-24:]
        if "region" in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    print(k, {c: "%.4g" % (x / cnt[(k, c)]) for c, x in sorted(d.items())})
PY
