#!/usr/bin/env python3
"""Developer probe: ONE 512-image batch alone on the GPU (the latency of a single step) under the region-stage variants and help settings."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch, numpy as np
lsd = importlib.import_module("linesegmentdetector-slam_amd")
maps = bench.load_maps()
n, size = 512, 2048
d = torch.from_numpy(bench.make_batch(maps, n, size)).cuda()
lines = torch.zeros((n, 1024, 10), dtype=torch.int64, device="cuda"); counts = torch.zeros(n, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for waves, help_ in ((8, -1), (4, -1), (4, 0), (4, 64)):
    ctx = lsd.Context(0)
    ctx.set_region_waves(waves); ctx.set_region_help(help_)
    ts = []
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.enqueue_device(d.data_ptr(), n, size, size, lines.data_ptr(), 1024, counts.data_ptr(), stream=s); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    st = [ctx.fetch(i, lsd.DBG_STATS, lsd.scaled_size(size, size)) for i in range(n)]
    cyc = np.array([x["cycles_total"] for x in st]) / 1e6
    print("waves %d help %3d: step %.1f ms (min of 3), region %.1f ms; cycles/image mean %.0fM max %.0fM; helped images %d, help evals %d" % (
        waves, help_, min(ts[1:]), ctx.timings()["region"], cyc.mean(), cyc.max(), sum(1 for x in st if x["help_exports"] > 0), sum(x["help_evals"] for x in st)))
    ctx.close()

# single images and a 64-image shard alone (the helper pool takes the CUs the batch leaves free)
for ids in ([187], [1], [0], list(range(64))):
    dd = torch.from_numpy(np.stack([bench.make_image(maps, i, size) for i in ids])).cuda()
    for help_ in (-1, 0):
        ctx = lsd.Context(0); ctx.set_region_help(help_)
        ts = []
        for rep in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ctx.enqueue_device(dd.data_ptr(), len(ids), size, size, lines.data_ptr(), 1024, counts.data_ptr(), stream=s); torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        print("images %s help %2d: %.1f ms" % (ids if len(ids) < 4 else "0..%d" % (len(ids) - 1), help_, min(ts[1:])))
        ctx.close()
