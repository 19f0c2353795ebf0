#!/usr/bin/env python3
"""Developer probe (make stats STATFLAGS=-DLSD_REGION_BATCHPROF, LSD_HIP_LIB=.../liblsdhip_stats.so): the time of a grow() batch by
segment, single bench images alone on 4 waves.   tools/batchprof.py id id ..."""
import importlib, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
lsd = importlib.import_module("linesegmentdetector-slam_amd")
maps = bench.load_maps(); ctx = lsd.Context(0)
ctx.set_region_waves(4)
for i in [int(a) for a in sys.argv[1:]] or [0, 187]:
    img = bench.make_image(maps, i, 2048)
    for rep in range(2): ctx.run(img.copy(), want_lineim=False)
    st = ctx.fetch(0, lsd.DBG_STATS, lsd.scaled_size(2048, 2048))
    b = max(st["batches"], 1)
    seg = {"between batches": st["cycles_sums"], "entry -> words read": st["cycles_rect"], "-> classify start (winner, sincos)": st["cycles_nfa"], "-> accepted": st["cycles_mark"], "-> worklist done": st["cycles_refine"], "tile fetches (inside the 2nd)": st["cycles_tiles"]}
    print("image %d: region %.1f ms, grow %.1f Mcycles over %d batches = %.0f cycles per batch, %.2f px per batch" % (i, ctx.timings()["region"], st["cycles_grow"] / 1e6, b, st["cycles_grow"] / b, st["grown_px"] / b))
    for k, v in seg.items():
        print("   %-40s %6.0f cycles per batch" % (k, v / b))
