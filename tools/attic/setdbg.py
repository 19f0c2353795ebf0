import importlib, sys, numpy as np
sys.path.insert(0, '.')
import bench
lsd = importlib.import_module("linesegmentdetector-slam_amd")
maps = bench.load_maps()
ctx = lsd.Context(0)
for i in (187, 16, 0):
    img = bench.make_image(maps, i, 2048)
    for waves in (8, 4):
        ctx.set_region_waves(waves)
        ctx.run(img.copy(), want_lineim=False)
        st = ctx.fetch(0, lsd.DBG_STATS, lsd.scaled_size(2048, 2048))
        print(i, waves, "cycles %.1fM" % (st["cycles_total"]/1e6), "set_answers", st["set_answers"], "sets_founded", st["sets_founded"], "grows", st["grow_calls"], "grown", st["grown_px"], ctx.timings()["region"])
