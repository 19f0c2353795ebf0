#!/usr/bin/env python3
"""Developer probe (needs a library built with the event trace, LSD_EVENTS=1): per-wave timeline of one image's region stage.
   tools/event_trace.py image_id [out.npy]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
lsd = importlib.import_module("linesegmentdetector-slam_amd")
maps = bench.load_maps(); ctx = lsd.Context(0); ctx.set_region_waves(8)
i = int(sys.argv[1])
img = bench.make_image(maps, i, 2048)
for rep in range(2): ctx.run(img.copy(), want_lineim=False)
print("region ms %.1f" % ctx.timings()["region"])
cap = 16384
a = np.zeros(8 * cap * 4, np.int32)
assert ctx.L.lsd_debug_fetch(ctx.h, 0, 100, a.ctypes.data, a.nbytes) == 0
ev = a.reshape(8, cap, 4)
np.save(sys.argv[2] if len(sys.argv) > 2 else os.path.join("gpurun_out", "events_%d.npy" % i), ev)
