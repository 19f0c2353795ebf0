#!/usr/bin/env python3
"""Developer experiment (library built with -DLSD_REGION_MILESTONES): how well does the time an image took for the first 1/16 .. 1/2 of
its seeds predict its total?"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
lsd = importlib.import_module("linesegmentdetector-slam_amd")
maps = bench.load_maps(); n = 512
ctx = lsd.Context(0)
d = torch.from_numpy(bench.make_batch(maps, n, 2048)).cuda()
lines = torch.zeros((n, 1024, 10), dtype=torch.int64, device="cuda"); counts = torch.zeros(n, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for rep in range(2):
    ctx.enqueue_device(d.data_ptr(), n, 2048, 2048, lines.data_ptr(), 1024, counts.data_ptr(), stream=s); torch.cuda.synchronize()
v = np.zeros((n, 48), np.int64)
for i in range(n): assert ctx.L.lsd_debug_fetch(ctx.h, i, lsd.DBG_STATS, v[i].ctypes.data, v[i].nbytes) == 0
T = v[:, 8].astype(float)
for m, frac in enumerate((1 / 16, 1 / 8, 1 / 4, 1 / 2)):
    t = v[:, 24 + m].astype(float)
    ok = t > 0
    rem = T[ok] - t[ok]
    print("cursor at %5.3f of the seeds: elapsed/total mean %.2f (p10 %.2f p90 %.2f); corr(elapsed, remaining) %.2f; remaining/elapsed: p10 %.2f median %.2f p90 %.2f" % (
        frac, (t[ok] / T[ok]).mean(), *np.percentile(t[ok] / T[ok], [10, 90]), np.corrcoef(t[ok], rem)[0, 1], *np.percentile(rem / t[ok], [10, 50, 90])))
