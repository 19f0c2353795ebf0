import importlib, os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/oracle") else os.getcwd())
import torch
from oracle import oracle
lsd = importlib.import_module("linesegmentdetector-slam_amd")
oracle.build()
ctx = lsd.Context(0)
def cases():
    yy, xx = np.mgrid[0:900, 0:1200]
    yield "ramp_x", np.clip(xx * 255.0 / 1199, 0, 255).astype(np.uint8)
    yield "ramp_diag", np.clip((xx + yy) * 255.0 / 2100, 0, 255).astype(np.uint8)
    r = np.hypot(xx - 600, yy - 450)
    yield "radial", np.clip(r * 255.0 / 750, 0, 255).astype(np.uint8)
    yield "cone_small", np.clip(255 - r * 2, 0, 255).astype(np.uint8)
    b = np.zeros((900, 1200), np.uint8)
    b[:450, :600] = (xx[:450, :600] * 255 // 600); b[450:, 600:] = (yy[450:, 600:] - 450) * 255 // 450
    b[:450, 600:] = 128; b[450:, :600] = ((xx[450:, :600] + yy[450:, :600]) % 256)
    yield "blocks", b
    rng = np.random.default_rng(3)
    yield "ramp_noise", np.clip(xx * 255.0 / 1199 + rng.normal(0, 6, xx.shape), 0, 255).astype(np.uint8)
    big = np.clip(np.mgrid[0:2048, 0:2048][1] * 255.0 / 2047, 0, 255).astype(np.uint8)
    yield "ramp_2048", big
def cases2():
    yy, xx = np.mgrid[0:2000, 0:1500]
    yield "saw_x", ((xx % 120) * 255 // 119).astype(np.uint8)                 # bands of ~36 x 600 scaled pixels, one angle each
    yield "saw_diag", (((xx + yy) % 170) * 255 // 169).astype(np.uint8)
    yy, xx = np.mgrid[0:7000, 0:400]
    yield "saw_tall", ((xx % 120) * 255 // 119).astype(np.uint8)              # > 65535 pixels per region
    rng = np.random.default_rng(4)
    yy, xx = np.mgrid[0:2000, 0:1500]
    yield "saw_noise", np.clip((xx % 120) * 255.0 / 119 + rng.normal(0, 3, xx.shape), 0, 255).astype(np.uint8)
import itertools
for name, img in itertools.chain(cases(), cases2()):
    for waves in (4, 8):
        ctx.set_region_waves(waves)
        t = time.time(); ref = oracle.lsd(img.copy(), debug=True); to = time.time() - t
        d = ref["dbg"]
        t = time.time(); lines, im = ctx.run(img.copy()); tg = time.time() - t
        used = (ctx.fetch(0, lsd.DBG_STATE, (d["w"], d["h"])) & 3).astype(np.uint8)
        st = ctx.fetch(0, lsd.DBG_STATS, (d["w"], d["h"]))
        ok = len(lines) == len(ref["lines"]) and np.array_equal(used, d["used"]) and np.array_equal(im, ref["lineIm"])
        if ok and len(lines):
            ok = all(np.abs(lines[f] - ref["lines"][f]).max() < 1e-6 for f in ("x1", "y1", "x2", "y2"))
        print(name, "waves", waves, "OK" if ok else "MISMATCH", "lines", len(lines), len(ref["lines"]), "used diff", int((used != d["used"]).sum()),
              "grows", st["grow_calls"], "grown", st["grown_px"], "oracle %.2fs gpu %.2fs" % (to, tg), flush=True)
