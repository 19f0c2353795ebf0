for rq in 1 0; do
  echo "== requeue $rq"
  LSD_REGION_REQUEUE=$rq HELPSTATS=1 timeout 120 python tools/hang_probe.py 512 2048 3 2>&1 | grep -v amdgpu.ids | grep "rep\|help:\|DIFF"
  LSD_REGION_REQUEUE=$rq LSD_REGION_HELP=0 LSD_HIP_LIB=$PWD/linesegmentdetector-slam_amd/liblsdhip_stats.so timeout 200 python tools/one_stats.py 8 45 77 110 1 0 187 2>&1 | grep -v amdgpu.ids | python -c "
import sys,re,ast
for l in sys.stdin:
    m=re.match(r'(\d+) waves 8 region ms ([\d.]+) (\{.*\})',l)
    if not m: print(l.strip()[:200]); continue
    d=ast.literal_eval(m.group(3))
    print('  ',m.group(1), m.group(2),'ms total',d['cycles_total'],'eval',d['cycles_eval'],'at cursor',d.get('cycles_eval_at_cursor'),'redos',d['spec_redos'],'requeued',d.get('requeued_ahead'),'discards',d['spec_discards'],'wait',d['cycles_wait'],'noslot',round(d['wait_noslot']/1e6,1),'commit',d['cycles_commit'])
"
done
