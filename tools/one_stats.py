import importlib, os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import bench, torch
lsd = importlib.import_module("linesegmentdetector-slam_amd")
maps = bench.load_maps(); ctx = lsd.Context(0)
ids = [int(a) for a in sys.argv[1:]] or [0, 27]
for i in ids:
    img = bench.make_image(maps, i, 2048)
    for rep in range(2): ctx.run(img.copy(), want_lineim=False)
    st = ctx.fetch(0, lsd.DBG_STATS, lsd.scaled_size(2048, 2048))
    print(i, ctx.timings()["region"], {k: (round(v / 1e6, 1) if k.startswith("cycles") else v) for k, v in st.items()})
