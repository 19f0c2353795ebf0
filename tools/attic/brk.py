#!/usr/bin/env python3
"""Developer probe: turns the lines of tools/one_stats.py (developer build) into a percentage breakdown of the wave time.
   ... tools/one_stats.py 8 110 | python tools/brk.py"""
import sys,re,ast
for l in sys.stdin:
    m=re.match(r'(\d+) waves (\d) region ms ([\d.]+) (\{.*\})',l)
    if not m: print(l.strip()[:200]); continue
    d=ast.literal_eval(m.group(4)); nw=int(m.group(2))
    tot=d['cycles_total']*nw
    g=lambda k: round(100*d.get(k,0)/tot,1)
    print(m.group(1),'w',nw, m.group(3),'ms total',d['cycles_total'],'M | % of wave time: eval',g('cycles_eval'),'(grow',g('cycles_grow'),'tiles',g('cycles_tiles'),'rect',g('cycles_rect'),'nfa',g('cycles_nfa'),') small',g('cycles_small'),'refill',g('cycles_refill'),'select',g('cycles_select'),'commit',g('cycles_commit'),'wait',g('cycles_wait'),'noslot',round(100*d['wait_noslot']/1e6/tot,1),'| batches',d['batches'],'grown',d['grown_px'],'grows',d['grow_calls'],'small steps',d['small_steps'],'redos',d.get('spec_redos'),'disc',d.get('spec_discards'),'requeued',d.get('requeued_ahead'),'exact',d['exact_angle_evals'])
