#!/usr/bin/env python3
"""Developer probe: where one host call on the reference's own map (map1, 400x350) spends its time."""
import importlib, os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
lsd = importlib.import_module("linesegmentdetector-slam_amd")
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "maps.npz"))
name = sys.argv[1] if len(sys.argv) > 1 else "map1"
img = z[name]
ctx = lsd.Context(0)
for want in (True, False):
    ts = []
    for i in range(60):
        m = img.copy()
        t = time.perf_counter(); lines, im = ctx.run(m, want_lineim=want); ts.append(time.perf_counter() - t)
    k = ctx.timings()
    print(name, img.shape, "lineim" if want else "no lineim", "wall median %.3f ms min %.3f" % (np.median(ts[10:]) * 1e3, np.min(ts) * 1e3),
          "kernels", {a: round(b, 3) for a, b in k.items()}, "lines", len(lines))
