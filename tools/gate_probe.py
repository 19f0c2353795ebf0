#!/usr/bin/env python3
"""Developer probe: one step of the bench batch at a time with help across workgroups, by the time an image must have run before it asks
(GATE, x 1024 clocks) and the helper wavefronts per image.   tools/gate_probe.py [n]"""
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
    for help_, gate, wb in ((0, 12000, 10), (24, 70000, 30), (24, 70000, 60), (24, 70000, 100), (24, 40000, 60), (24, 40000, 100), (24, 12000, 100), (48, 40000, 100)):
        ctx = lsd.Context(0)
        ctx.set_region_waves(waves); ctx.set_region_help(help_); ctx.reserve(n, size, size)
        ctx.debug_set_tuning("GATE", gate); ctx.debug_set_tuning("WB", wb)
        ts = []
        for rep in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ctx.enqueue_device(d.data_ptr(), n, size, size, out.data_ptr(), 1024, cnt.data_ptr())
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        st = [ctx.fetch(i, lsd.DBG_STATS, lsd.scaled_size(size, size)) for i in range(n)]
        print("waves %d help %3d gate %6d (%.0f ms) wb %3d: %.1f ms per step (min %.1f); lines %d; helped evaluations %d, images that exported %d" % (waves, help_, gate, gate * 1024 / 2.1e6, wb, float(np.median(ts[1:])), min(ts[1:]), int(cnt.sum()), sum(x["help_evals"] for x in st), sum(1 for x in st if x["help_exports"] > 0)), flush=True)
        del ctx
