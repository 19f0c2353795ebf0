#!/bin/bash
# developer probe: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes as MI355X_MICROARCH.md prescribes) of a command's kernels
#   tools/pmc_hbm.sh <tag> <python script> [args...]      (run on the GPU box)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
for cnt in FETCH_SIZE WRITE_SIZE; do
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_${cnt}_$tag
  rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d $out -o pmc -- python3 $GRAFT_REPO_ROOT/"$@" > $out.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/pmc_to_traffic.py $GRAFT_REPO_ROOT/gpurun_out/pmc_FETCH_SIZE_$tag $GRAFT_REPO_ROOT/gpurun_out/pmc_WRITE_SIZE_$tag $GRAFT_REPO_ROOT/gpurun_out/traffic_$tag.json | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d['kernels'].items(): print(k, 'fetch %.2f GB write %.2f GB total %.2f GB' % (2*v['FETCH_SIZE_KB']*1024/1e9, v['WRITE_SIZE_KB']*1024/1e9, v['hbm_bytes_per_launch']/1e9))
"
