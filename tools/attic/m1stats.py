import importlib, os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import torch
lsd = importlib.import_module("linesegmentdetector-slam_amd")
z = np.load("tests/golden/maps.npz")
ctx = lsd.Context(0)
for name in ("map1", "aisle1"):
    img = z[name]
    for waves in (4, 8):
        ctx.set_region_waves(waves)
        for rep in range(3): ctx.run(img.copy(), want_lineim=False)
        st = ctx.fetch(0, lsd.DBG_STATS, lsd.scaled_size(img.shape[1], img.shape[0]))
        print(name, "waves", waves, "region ms %.3f" % ctx.timings()["region"], {k: (round(v / 1e6, 2) if k.startswith("cycles") else v) for k, v in st.items() if v})
