import importlib, sys, numpy as np
sys.path.insert(0, '.')
import bench
from oracle import oracle
lsd = importlib.import_module("linesegmentdetector-slam_amd")
maps = bench.load_maps()
ctx = lsd.Context(0)
n, size, first = 48, 1024, 0
batch = bench.make_batch(maps, n, size, first)
refs = {i: oracle.lsd(batch[i].copy(), debug=True) for i in range(n)}
wh = lsd.scaled_size(size, size)
nbad = 0
for rep in range(30):
    ctx.set_region_waves(4 if rep % 2 else 0)
    lines, offs, ims = ctx.run_batch(batch.copy())
    for i in range(n):
        if i == 13: continue
        ref = refs[i]
        if offs[i + 1] - offs[i] != len(ref["lines"]) or not np.array_equal(ims[i], ref["lineIm"]):
            used = (ctx.fetch(i, lsd.DBG_STATE, wh) & 3).astype(np.uint8)
            st = ctx.fetch(i, lsd.DBG_STATS, wh)
            ys, xs = np.nonzero(used != ref["dbg"]["used"])
            print("rep", rep, "waves", 4 if rep % 2 else 8, "image", i, "lines", int(offs[i+1]-offs[i]), "vs", len(ref["lines"]), "usedMap diff", len(ys), "bbox", (xs.min(), xs.max(), ys.min(), ys.max()) if len(ys) else None,
                  "gpu codes there", np.unique(used[ys, xs]).tolist(), "ref codes", np.unique(ref["dbg"]["used"][ys, xs]).tolist(), "set answers", st.get("set_answers"), "founded", st.get("sets_founded"), "redos", st["spec_redos"], flush=True)
            nbad += 1
print("bad", nbad)
