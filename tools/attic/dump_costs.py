#!/usr/bin/env python3
"""Developer probe: per-image cycles of the region stage with 4 and with 8 waves (whole bench batch) next to nb -> gpurun_out/costs.npz"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, torch
lsd = importlib.import_module("linesegmentdetector-slam_amd")
maps = bench.load_maps()
n, size = 512, 2048
ctx = lsd.Context(0)
d = torch.from_numpy(bench.make_batch(maps, n, size)).cuda()
lines = torch.zeros((n, 1024, 10), dtype=torch.int64, device="cuda"); counts = torch.zeros(n, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
wh = lsd.scaled_size(size, size)
res = {}
for waves in (4, 8):
    ctx.set_region_waves(waves)
    for rep in range(2):
        ctx.enqueue_device(d.data_ptr(), n, size, size, lines.data_ptr(), 1024, counts.data_ptr(), stream=s); torch.cuda.synchronize()
    res["t%d" % waves] = np.array([ctx.fetch(i, lsd.DBG_STATS, wh)["cycles_total"] for i in range(n)])
    print(waves, ctx.timings()["region"])
res["nb"] = np.array([ctx.fetch(i, lsd.DBG_NB, wh) for i in range(n)])
np.savez(os.path.join(ROOT, "gpurun_out", "costs.npz"), **res)
