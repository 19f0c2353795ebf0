import importlib, sys, numpy as np
sys.path.insert(0, '.')
import bench
from oracle import oracle
lsd = importlib.import_module("linesegmentdetector-slam_amd")
maps = bench.load_maps()
ctx = lsd.Context(0)
n, size, first = 48, 1024, 0
batch = bench.make_batch(maps, n, size, first)
refs = [oracle.lsd(batch[i].copy(), debug=True) for i in range(n)]
wh = lsd.scaled_size(size, size)
for rep in range(6):
    ctx.set_region_waves(4 if rep % 2 else 0)
    lines, offs, ims = ctx.run_batch(batch.copy())
    bad = []
    for i in range(n):
        used = (ctx.fetch(i, lsd.DBG_STATE, wh) & 3).astype(np.uint8)
        st = ctx.fetch(i, lsd.DBG_STATS, wh)
        ok = offs[i + 1] - offs[i] == len(refs[i]["lines"]) and np.array_equal(used, refs[i]["dbg"]["used"]) and np.array_equal(ims[i], refs[i]["lineIm"])
        if not ok:
            bad.append((i, int(offs[i + 1] - offs[i]), len(refs[i]["lines"]), int((used != refs[i]["dbg"]["used"]).sum()), st["set_answers"], st["sets_founded"]))
    print("rep", rep, "bad images (i, lines, ref lines, usedMap diff, set answers, sets founded):", bad, flush=True)
