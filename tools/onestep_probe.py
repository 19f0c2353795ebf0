#!/usr/bin/env python3
"""Developer probe: one step of the bench batch at a time, by region-stage variant (waves per image x help across workgroups).
   tools/onestep_probe.py [n]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, torch
lsd = importlib.import_module("linesegmentdetector-slam_amd")
n, size = int(sys.argv[1]) if len(sys.argv) > 1 else 512, 2048
maps = bench.load_maps()
d = torch.from_numpy(bench.make_batch(maps, n, size)).cuda()
out = torch.zeros((n, 1024, 10), dtype=torch.int64, device="cuda"); cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
for waves in (8, 4):
    for help_ in [int(x) for x in os.environ.get("HELPS", "-1,0,8,24,48").split(",")]:
        for hist in [int(x) for x in os.environ.get("HISTS", "0,1").split(",")]:
            ctx = lsd.Context(0)
            ctx.set_region_waves(waves); ctx.set_region_help(help_); ctx.reserve(n, size, size)
            ctx.set_cost_history(hist)
            ts = []
            for rep in range(6):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                ctx.enqueue_device(d.data_ptr(), n, size, size, out.data_ptr(), 1024, cnt.data_ptr())
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            print("waves %d help %3d history %d: %.1f ms per step (min %.1f), region %.1f; lines %d" % (waves, help_, hist, float(np.median(ts[1:])), min(ts[1:]), ctx.timings()["region"], int(cnt.sum())), flush=True)
            del ctx
