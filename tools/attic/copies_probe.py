#!/usr/bin/env python3
"""Developer probe: one bench image alone against 256 copies of it in one launch (every CU busy with the same work): what the
region stage loses to contention between compute units (shared L2 / instruction fetch).   tools/copies_probe.py id [id ...]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
lsd = importlib.import_module("linesegmentdetector-slam_amd")
maps = bench.load_maps(); ctx = lsd.Context(0)
s = torch.cuda.current_stream().cuda_stream
wh = lsd.scaled_size(2048, 2048)
for i in [int(a) for a in sys.argv[1:]] or [0]:
    img = bench.make_image(maps, i, 2048)
    for n in (1, 256):
        d = torch.from_numpy(np.broadcast_to(img, (n, 2048, 2048)).copy()).cuda()
        lines = torch.zeros((n, 1024, 10), dtype=torch.int64, device="cuda"); counts = torch.zeros(n, dtype=torch.int32, device="cuda")
        best = 1e9
        for rep in range(3):
            ctx.enqueue_device(d.data_ptr(), n, 2048, 2048, lines.data_ptr(), 1024, counts.data_ptr(), stream=s); torch.cuda.synchronize()
            best = min(best, ctx.timings()["region"])
        def total(j):                                       # (slot 8 of the stats record; developer builds have 48 words, the product 32)
            for words in (32, 48):
                a = np.zeros(words, np.int64)
                if ctx.L.lsd_debug_fetch(ctx.h, j, lsd.DBG_STATS, a.ctypes.data, a.nbytes) == 0:
                    return a[8]
            raise RuntimeError("stats fetch failed")
        cyc = np.array([total(j) for j in range(0, n, max(1, n // 16))]) / 1e6
        print("image", i, "copies", n, "region ms %.1f" % best, "Mcycles per image: mean %.1f max %.1f" % (cyc.mean(), cyc.max()), flush=True)
        del d, lines, counts
