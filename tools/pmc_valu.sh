#!/bin/bash
# developer probe: how busy the vector unit is with three 4-wave workgroups per CU -- SQ counters of ONE launch of 768 copies of a bench image
#   tools/pmc_valu.sh <tag> [image]
tag=$1; img=${2:-1}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_valu_${tag}
COPIES=768 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES \
  --kernel-trace --output-format csv -d $out -o pmc -- python3 $GRAFT_REPO_ROOT/tools/contention_curve.py $img > $out.log 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(list)
for p in glob.glob("$out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "k_region" not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("(")[0][-20:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, d in agg.items():
    v = {c: x / cnt[(k, c)] for c, x in d.items()}
    ms = sum(dur[k]) / len(dur[k]) / 1e6
    print(k, "per launch:", {c: "%.3g" % x for c, x in v.items()}, "launch %.1f ms (serialised by the counter pass)" % ms)
    # SQ_ACTIVE_INST_VALU counts cycles (x4: MI355X_MICROARCH.md) in which a SIMD's vector unit executes; 256 CUs x 4 SIMDs
    if "SQ_ACTIVE_INST_VALU" in v and "SQ_BUSY_CYCLES" in v:
        print("   VALU wave-instructions per wave-cycle %.3f; SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES %.3f; / SQ_BUSY_CYCLES %.3f" % (
            v["SQ_INSTS_VALU"] / v["SQ_WAVE_CYCLES"], v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"], v["SQ_ACTIVE_INST_VALU"] / v["SQ_BUSY_CYCLES"]))
PY
grep copies $out.log
