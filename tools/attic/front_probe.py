#!/usr/bin/env python3
"""Developer probe: average device time of the front-end kernels (K1 gauss, K2 gradient, K3 sort) over the bench batch."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, torch
lsd = importlib.import_module("linesegmentdetector-slam_amd")
maps = bench.load_maps()
n, size, reps = 512, 2048, int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = lsd.Context(0)
d = torch.from_numpy(bench.make_batch(maps, n, size)).cuda()
lines = torch.zeros((n, 1024, 10), dtype=torch.int64, device="cuda"); counts = torch.zeros(n, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
ctx.set_stop_after(lsd.STAGE_SORT)
acc = {}
for rep in range(reps + 2):
    ctx.enqueue_device(d.data_ptr(), n, size, size, lines.data_ptr(), 1024, counts.data_ptr(), stream=s); torch.cuda.synchronize()
    if rep >= 2:
        for k, v in ctx.timings().items(): acc.setdefault(k, []).append(v)
w, h = lsd.scaled_size(size, size)
for k in ("gauss", "gradient", "sort"):
    a = np.array(acc[k]); print("%-9s mean %.3f ms  min %.3f  max %.3f" % (k, a.mean(), a.min(), a.max()))
g = np.array(acc["gradient"]).mean()
print("gradient: %.0f GB/s algorithmic (25 B x %d px x %d) = %.3f of 8 TB/s" % (25.0 * w * h * n / g / 1e6, w * h, n, 25.0 * w * h * n / g / 1e6 / 8000))
