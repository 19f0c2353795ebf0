#!/usr/bin/env python3
"""Developer probe: per-image cost of the region stage on the bench batch (8 waves, one step alone) next to what is known before the
stage starts (K3's count of candidate pixels), as CSV.   tools/image_costs.py out.csv"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, torch
lsd = importlib.import_module("linesegmentdetector-slam_amd")
n, size = 512, 2048
maps = bench.load_maps()
d = torch.from_numpy(bench.make_batch(maps, n, size)).cuda()
ctx = lsd.Context(0)
ctx.set_region_waves(8); ctx.set_region_help(0); ctx.reserve(n, size, size)
out = torch.zeros((n, 1024, 10), dtype=torch.int64, device="cuda"); cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
for rep in range(2):
    ctx.enqueue_device(d.data_ptr(), n, size, size, out.data_ptr(), 1024, cnt.data_ptr()); torch.cuda.synchronize()
wh = lsd.scaled_size(size, size)
with open(sys.argv[1], "w") as f:
    f.write("image,nb,nseed,cycles,grown_px,grow_calls,nfa_calls,seeds,lines,set_answers\n")
    for i in range(n):
        st = ctx.fetch(i, lsd.DBG_STATS, wh)
        f.write("%d,%d,%d,%d,%d,%d,%d,%d,%d,%d\n" % (i, ctx.fetch(i, lsd.DBG_NB, wh), ctx.fetch(i, lsd.DBG_NSEED, wh), st["cycles_total"], st["grown_px"], st["grow_calls"], st["nfa_calls"], st["seeds"], int(cnt[i]), st["set_answers"]))
print("written", sys.argv[1])
