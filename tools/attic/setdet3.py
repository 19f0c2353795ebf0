import importlib, sys, numpy as np
sys.path.insert(0, '.')
import bench
lsd = importlib.import_module("linesegmentdetector-slam_amd")
maps = bench.load_maps()
ctx = lsd.Context(0)
n, size, first = 48, 1024, 0
batch = bench.make_batch(maps, n, size, first)
ref = None
for rep in range(8):
    ctx.set_region_waves(4 if rep % 2 else 0)
    lines, offs, ims = ctx.run_batch(batch.copy())
    if ref is None:
        ref = (lines.copy(), offs.copy(), ims.copy()); continue
    if not np.array_equal(offs, ref[1]): print("rep", rep, "offsets differ")
    if not np.array_equal(ims, ref[2]): print("rep", rep, "lineIm differs in", int((ims != ref[2]).sum()))
    if lines.tobytes() != ref[0].tobytes():
        for i in range(n):
            a, b = lines[offs[i]:offs[i+1]], ref[0][ref[1][i]:ref[1][i+1]]
            if a.tobytes() != b.tobytes():
                for f in a.dtype.names:
                    if not np.array_equal(a[f], b[f], equal_nan=True):
                        j = np.nonzero(~((a[f] == b[f]) | (np.isnan(a[f].astype(float)) & np.isnan(b[f].astype(float)))))[0]
                        print("rep", rep, "image", i, "field", f, "rows", j[:5], a[f][j[:3]], b[f][j[:3]])
    else:
        print("rep", rep, "identical")
