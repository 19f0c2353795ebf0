#!/usr/bin/env python3
"""Developer probe: the front end's kernel times for the library in LSD_HIP_LIB on the bench batch (one step at a time), with a digest of
the Gaussian images so that variants of K1 can be told to be bit-identical.   LSD_HIP_LIB=... tools/k1_probe.py [n]"""
import hashlib, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, torch
lsd = importlib.import_module("linesegmentdetector-slam_amd")
n, size = int(sys.argv[1]) if len(sys.argv) > 1 else 512, 2048
maps = bench.load_maps()
d = torch.from_numpy(bench.make_batch(maps, n, size)).cuda()
ctx = lsd.Context(0)
ctx.reserve(n, size, size)
out = torch.zeros((n, 1024, 10), dtype=torch.int64, device="cuda"); cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
lim = torch.zeros((n, size, size), dtype=torch.uint8, device="cuda")
acc = {}
for rep in range(6):
    ctx.enqueue_device(d.data_ptr(), n, size, size, out.data_ptr(), 1024, cnt.data_ptr(), d_line_ims=lim.data_ptr())
    torch.cuda.synchronize()
    if rep:
        for k, v in ctx.timings().items(): acc.setdefault(k, []).append(v)
wh = lsd.scaled_size(size, size)
h = hashlib.sha1()
for i in range(0, n, max(1, n // 16)): h.update(ctx.fetch(i, lsd.DBG_GAUSS, wh).tobytes())
print(os.environ.get("LSD_HIP_LIB", "default"), {k: round(float(np.median(v)), 3) for k, v in acc.items()}, "gauss digest", h.hexdigest()[:12], "lines", int(cnt.sum()))
