import importlib, sys, numpy as np, hashlib
sys.path.insert(0, '.')
import bench
lsd = importlib.import_module("linesegmentdetector-slam_amd")
maps = bench.load_maps()
ctx = lsd.Context(0)
for (n, size, first) in ((48, 1024, 0), (96, 1024, 140), (32, 2048, 180)):
    batch = bench.make_batch(maps, n, size, first)
    ref = None; bad = 0
    for rep in range(12):
        for waves in (0, 4):
            ctx.set_region_waves(waves)
            lines, offs, ims = ctx.run_batch(batch.copy())
            cur = hashlib.md5(lines.tobytes() + offs.tobytes() + ims.tobytes()).hexdigest()
            if ref is None: ref = cur
            if cur != ref:
                bad += 1
    st = [ctx.fetch(i, lsd.DBG_STATS, lsd.scaled_size(size, size)) for i in range(n)]
    print(n, size, first, "runs 24, differing", bad, "set answers", sum(x["set_answers"] for x in st), "founded", sum(x["sets_founded"] for x in st), flush=True)
