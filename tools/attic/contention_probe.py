#!/usr/bin/env python3
"""Developer probe: the SAME image replicated n times (so every workgroup has identical work): region-stage time vs n shows what
sharing the device costs a single image's latency chain."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, torch
lsd = importlib.import_module("linesegmentdetector-slam_amd")
maps = bench.load_maps()
size = 2048
which = int(sys.argv[1]) if len(sys.argv) > 1 else 27
img = bench.make_image(maps, which, size)
ctx = lsd.Context(0)
nmax = 512
d = torch.from_numpy(np.broadcast_to(img, (nmax, size, size)).copy()).cuda()
lines = torch.zeros((nmax, 1024, 10), dtype=torch.int64, device="cuda"); counts = torch.zeros(nmax, dtype=torch.int32, device="cuda")
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
for waves in (4, 8):
    ctx.set_region_waves(waves)
    for n in (1, 16, 64, 128, 256, 512):
        if waves == 8 and n > 256:
            continue
        for rep in range(2):
            ctx.enqueue_device(d.data_ptr(), n, size, size, lines.data_ptr(), 1024, counts.data_ptr(), stream=st.cuda_stream)
            torch.cuda.synchronize()
        t = ctx.timings()
        print("waves %d n %3d region %.1f ms" % (waves, n, t["region"]), flush=True)
