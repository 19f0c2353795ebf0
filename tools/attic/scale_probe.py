#!/usr/bin/env python3
"""Developer probe: region-stage time vs number of images in flight."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import torch
lsd = importlib.import_module("linesegmentdetector-slam_amd")
maps = bench.load_maps()
ctx = lsd.Context(0)
size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ns = [int(x) for x in sys.argv[2:]] or [1, 8, 64, 256, 512]
nmax = max(ns)
host = bench.make_batch(maps, nmax, size)
d = torch.from_numpy(host).cuda()
lines = torch.zeros((nmax, 1024, 10), dtype=torch.int64, device="cuda")
counts = torch.zeros(nmax, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for n in ns:
    for rep in range(2):
        ctx.enqueue_device(d.data_ptr(), n, size, size, lines.data_ptr(), 1024, counts.data_ptr(), stream=s)
        torch.cuda.synchronize()
    t = ctx.timings()
    st = ctx.fetch(0, lsd.DBG_STATS, lsd.scaled_size(size, size))
    print(n, {k: round(v, 3) for k, v in t.items()}, int(counts[:n].sum()), flush=True)
    if n == ns[-1]:
        for i in range(min(n, 6)):
            st = ctx.fetch(i, lsd.DBG_STATS, lsd.scaled_size(size, size))
            print(i, int(counts[i]), {k: (v // 1000 if k.startswith('cycles') else v) for k, v in st.items()}, flush=True)
