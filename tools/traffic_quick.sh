R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for cnt in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $cnt --kernel-trace --output-format csv -d $O/r04k_pmc_$cnt -o pmc -- python3 $R/bench.py --steps 8 --warmup 8 --no-cpu-baseline > $O/r04k_pmc_$cnt.log 2>&1
done
python3 $R/tools/pmc_to_traffic.py $O/r04k_pmc_FETCH_SIZE $O/r04k_pmc_WRITE_SIZE $O/r04k_traffic.json r04k | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d['kernels'].items(): print(k, 'fetch %.2f GB write %.2f GB total %.2f GB' % (2*v['FETCH_SIZE_KB']*1024/1e9, v['WRITE_SIZE_KB']*1024/1e9, v['hbm_bytes_per_launch']/1e9), v['launches'])
"
