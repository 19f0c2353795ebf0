#!/usr/bin/env python3
import importlib, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import oracle
lsd = importlib.import_module("linesegmentdetector-slam_amd")
oracle.build()
maps = np.load(os.path.join(ROOT, "tests", "golden", "maps.npz"))
for name in ("f3key", "f4key", "aisle1", "mapValue"):
    img = maps[name]
    ref = oracle.lsd(img.copy(), debug=True); d = ref["dbg"]
    for waves in (8, 4):
        for help_ in (-1, 0):
            ctx = lsd.Context(0); ctx.set_region_waves(waves); ctx.set_region_help(help_)
            res = []
            for rep in range(40):
                lines, im = ctx.run(img.copy())
                used = (ctx.fetch(0, lsd.DBG_STATE, (d["w"], d["h"])) & 3).astype(np.uint8)
                st = ctx.fetch(0, lsd.DBG_STATS, (d["w"], d["h"]))
                res.append((len(lines), int((used != d["used"]).sum())))
            print(name, "waves", waves, "help", help_, "ref lines", len(ref["lines"]), "bad runs of 40:", [r for r in res if r != (len(ref["lines"]), 0)], flush=True)
            ctx.close()
