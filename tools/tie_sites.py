#!/usr/bin/env python3
"""Developer probe: WHERE an image's near-ties are (lsd_last_sensitivity).  Needs the experiment build with -DLSD_TIE_SITES
(make -C linesegmentdetector-slam_amd/csrc exp EXPFLAGS=-DLSD_TIE_SITES EXPFLAGS8=-DLSD_TIE_SITES; LSD_HIP_LIB=.../liblsdhip_exp.so):
the counter then is a decimal record, three digits per site: grow test, orientation flip, density, distances (Refiner / Reducer),
rectangle edges, aligned count, NFA margins.     tools/tie_sites.py [campaign image numbers ...]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: F401
lsd = importlib.import_module("linesegmentdetector-slam_amd")
from campaign_images import campaign_image
ctx = lsd.Context(0)
SITES = ("grow", "flip", "dens", "dist", "edge", "align", "nfa")


def show(name, img, kw=None):
    ctx.run(img.copy(), lsd.make_params(**kw) if kw else None)
    w, h = lsd.scaled_size(img.shape[1], img.shape[0])
    v = ctx.fetch(0, lsd.DBG_STATS, (w, h))["near_ties"]
    print(name, img.shape, {s: (v // 1000 ** k) % 1000 for k, s in enumerate(SITES)}, flush=True)


z = np.load(os.path.join(ROOT, "tests", "golden", "maps.npz"))
for k in ("map1", "mapValue", "aisle1", "aisle2", "aisle3", "f3key", "f4key"):
    show(k, z[k])
for i in [int(x) for x in sys.argv[1:]] or list(range(12)) + [1759, 2995]:
    img, kw, _ = campaign_image(i, False)
    show("campaign %d" % i, img, kw)
