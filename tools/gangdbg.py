import importlib, os, sys, numpy as np
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/oracle") else os.getcwd())
from oracle import oracle
lsd = importlib.import_module("linesegmentdetector-slam_amd")
oracle.build()
d = np.load("tests/golden/maps.npz")
img = d["mapValue"]
ref = oracle.lsd(img.copy(), debug=True)
for waves in (8, 4):
    ctx = lsd.Context(0); ctx.set_region_waves(waves)
    lines, im = ctx.run(img.copy())
    wh = (ref["dbg"]["w"], ref["dbg"]["h"])
    used = (ctx.fetch(0, lsd.DBG_STATE, wh) & 3).astype(np.uint8)
    print("waves", waves, "lines", len(lines), len(ref["lines"]), "used diff", int((used != ref["dbg"]["used"]).sum()), "lineIm equal", np.array_equal(im, ref["lineIm"]))
