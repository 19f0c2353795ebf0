#!/usr/bin/env python3
"""Developer probe: region-stage time of the bench batch with 4 and with 8 wavefronts per image; per-image cycles of a few heavy images."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, torch
lsd = importlib.import_module("linesegmentdetector-slam_amd")
maps = bench.load_maps()
n, size = int(sys.argv[1]) if len(sys.argv) > 1 else 512, 2048
ctx = lsd.Context(0)
d = torch.from_numpy(bench.make_batch(maps, n, size)).cuda()
lines = torch.zeros((n, 1024, 10), dtype=torch.int64, device="cuda"); counts = torch.zeros(n, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
wh = lsd.scaled_size(size, size)
for waves in (4, 8):
    ctx.set_region_waves(waves)
    for rep in range(2):
        ctx.enqueue_device(d.data_ptr(), n, size, size, lines.data_ptr(), 1024, counts.data_ptr(), stream=s); torch.cuda.synchronize()
    tot = np.array([ctx.fetch(i, lsd.DBG_STATS, wh)["cycles_total"] for i in range(n)]) / 1e6
    print("waves", waves, {k: round(v, 2) for k, v in ctx.timings().items()}, "Mcycles mean %.0f p90 %.0f max %.0f" % (tot.mean(), np.percentile(tot, 90), tot.max()), flush=True)
ctx.set_region_waves(0)
for i in (187, 355, 0, 1):
    img = torch.from_numpy(bench.make_image(maps, i, size)).cuda()
    for waves in (4, 8):
        ctx.set_region_waves(waves)
        for rep in range(2):
            ctx.enqueue_device(img.data_ptr(), 1, size, size, lines.data_ptr(), 1024, counts.data_ptr(), stream=s); torch.cuda.synchronize()
        print("image", i, "alone, waves", waves, "region ms %.2f" % ctx.timings()["region"], flush=True)
