#!/usr/bin/env python3
"""Developer probe: n copies of one bench image in one launch of the 4-wave region stage (help off): region time and clocks per image
against the number of workgroups resident per CU (256 copies: one per CU, 768: three).  What a workgroup loses to its neighbours.
   tools/contention_curve.py [image ...]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
lsd = importlib.import_module("linesegmentdetector-slam_amd")
maps = bench.load_maps(); ctx = lsd.Context(0)
ctx.set_region_waves(4); ctx.set_region_help(0)
s = torch.cuda.current_stream().cuda_stream
for i in [int(a) for a in sys.argv[1:]] or [1]:
    img = bench.make_image(maps, i, 2048)
    for n in [int(x) for x in os.environ.get("COPIES", "1,64,256,512,768,1536").split(",")]:
        d = torch.from_numpy(np.broadcast_to(img, (n, 2048, 2048)).copy()).cuda()
        lines = torch.zeros((n, 1024, 10), dtype=torch.int64, device="cuda"); counts = torch.zeros(n, dtype=torch.int32, device="cuda")
        best = 1e9
        for rep in range(3):
            ctx.enqueue_device(d.data_ptr(), n, 2048, 2048, lines.data_ptr(), 1024, counts.data_ptr(), stream=s); torch.cuda.synchronize()
            best = min(best, ctx.timings()["region"])
        cyc = ctx.fetch_stats_block(n)[:, 8] / 1e6
        print("image %d copies %4d: region %.1f ms = %.2f images per ms; Mcycles per image mean %.1f max %.1f" % (i, n, best, n / best, cyc.mean(), cyc.max()), flush=True)
        del d, lines, counts
