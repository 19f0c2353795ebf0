#!/usr/bin/env python3
"""Developer probe: region-stage counters of single bench images run alone (use LSD_HIP_LIB=.../liblsdhip_stats.so for the stopwatches).
   tools/one_stats.py [waves] id id ..."""
import importlib, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
lsd = importlib.import_module("linesegmentdetector-slam_amd")
maps = bench.load_maps(); ctx = lsd.Context(0)
args = [int(a) for a in sys.argv[1:]]
waves = args[0] if args and args[0] in (4, 8) else 0
ids = args[1:] if waves else (args or [0, 27])
ctx.set_region_waves(waves)
for i in ids:
    img = bench.make_image(maps, i, 2048)
    for rep in range(2): ctx.run(img.copy(), want_lineim=False)
    st = ctx.fetch(0, lsd.DBG_STATS, lsd.scaled_size(2048, 2048))
    print(i, "waves", waves, "region ms %.1f" % ctx.timings()["region"], {k: (round(v / 1e6, 1) if k.startswith("cycles") else v) for k, v in st.items() if v})
