#!/usr/bin/env python3
"""Developer probe for the help across workgroups in the region stage: one heavy bench image in a launch with N - 1 copies of a light
one (the CUs that finish early become helpers).   tools/help_probe.py heavy light n"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
lsd = importlib.import_module("linesegmentdetector-slam_amd")
maps = bench.load_maps(); ctx = lsd.Context(0)
s = torch.cuda.current_stream().cuda_stream
wh = lsd.scaled_size(2048, 2048)
heavy, light, n = (int(a) for a in sys.argv[1:4])
a = np.empty((n, 2048, 2048), np.uint8)
a[:] = bench.make_image(maps, light, 2048)
a[0] = bench.make_image(maps, heavy, 2048)
d = torch.from_numpy(a).cuda()
lines = torch.zeros((n, 1024, 10), dtype=torch.int64, device="cuda"); counts = torch.zeros(n, dtype=torch.int32, device="cuda")
for rep in range(3):
    ctx.enqueue_device(d.data_ptr(), n, 2048, 2048, lines.data_ptr(), 1024, counts.data_ptr(), stream=s); torch.cuda.synchronize()
    st = ctx.fetch(0, lsd.DBG_STATS, wh)
    ev = sum(ctx.fetch(i, lsd.DBG_STATS, wh)["help_evals"] for i in range(n))
    print("region %.1f ms | heavy image: %.1f Mcycles, grow calls %d, exports %d, redos %d discards %d | evaluations by helpers %d | lines %d" % (
        ctx.timings()["region"], st["cycles_total"] / 1e6, st["grow_calls"], st["help_exports"], st["spec_redos"], st["spec_discards"], ev, int(counts[0])), flush=True)
