#!/usr/bin/env python3
"""Developer tool: randomized parity campaign in BATCH mode -- groups of random synthetic maps of one size (walls, noise, random
parameters per group) run as one batch with the help across workgroups on (lsd_set_region_help 24), both region-stage variants, and
compared image by image with the oracle.  Exercises what the single-image campaign (tools/campaign.py) cannot: helpers from finished
workgroups, several images' commit machinery at once.     tools/campaign_batch.py [groups] [images per group]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (before the HIP library)
from oracle import oracle
lsd = importlib.import_module("linesegmentdetector-slam_amd")
oracle.build()
ctx = lsd.Context(0)
ctx.set_region_help(24)                                    # (off by default since round 6: this campaign is about the help protocol)
groups = int(sys.argv[1]) if len(sys.argv) > 1 else 40
per = int(sys.argv[2]) if len(sys.argv) > 2 else 48


def synth(rng, rows, cols, heavy):
    m = np.zeros((rows, cols), np.uint8)
    m[rng.random((rows, cols)) < rng.uniform(0.0, 0.5)] = 255
    for _ in range(int(rng.integers(3, 200 if heavy else 30))):
        x0, y0 = rng.integers(2, cols - 2), rng.integers(2, rows - 2)
        L = int(rng.integers(10, 1500 if heavy else 300)); a = rng.choice([0, np.pi / 2, np.pi / 4, rng.uniform(0, np.pi)])
        t = np.arange(L)
        xs = np.clip((x0 + t * np.cos(a)).astype(int), 0, cols - 1); ys = np.clip((y0 + t * np.sin(a)).astype(int), 0, rows - 1)
        m[ys, xs] = 1
        if rng.random() < 0.3:
            m[np.clip(ys + 1, 0, rows - 1), xs] = 1
    if rng.random() < 0.3:
        m[rng.random((rows, cols)) < 0.01] = 1
    return m


bad = tot = helped = 0
t0 = time.time()
for gi in range(groups):
    rng = np.random.default_rng(77_000 + gi)
    rows, cols = int(rng.integers(300, 1600)), int(rng.integers(300, 1600))
    kw = {}
    if rng.random() < 0.3:
        kw = dict(sca=0.3, sig=float(rng.choice([0.6, 0.8])), angThre=float(rng.choice([22.5, 20.0, 30.0])), denThre=float(rng.choice([0.7, 0.6])), pseBin=int(rng.choice([1024, 512])))
    # one or two heavy images among light ones: the light ones finish first and their wavefronts help the heavy ones
    batch = np.stack([synth(rng, rows, cols, heavy=(k < 2)) for k in range(per)])
    ctx.set_region_waves(int(rng.choice([4, 8])))
    lines, offs, ims = ctx.run_batch(batch.copy(), lsd.make_params(**kw) if kw else None)
    wh = lsd.scaled_size(cols, rows)
    helped += sum(ctx.fetch(i, lsd.DBG_STATS, wh)["help_exports"] for i in range(per))
    for i in range(per):
        ref = oracle.lsd(batch[i].copy(), **kw)
        L = lines[offs[i]:offs[i + 1]]
        ok = len(L) == len(ref["lines"]) and np.array_equal(ims[i], ref["lineIm"])
        if ok and len(L):
            ok = all(np.abs(L[f] - ref["lines"][f]).max() < 1e-6 for f in ("x1", "y1", "x2", "y2")) and np.array_equal(L["orient"], ref["lines"]["orient"])
        tot += 1
        if not ok:
            bad += 1
            rc = oracle.lsd(batch[i].copy(), _lib=oracle.lib_cr(), **kw)
            eq = len(L) == len(rc["lines"]) and np.array_equal(ims[i], rc["lineIm"])
            print("MISMATCH group", gi, "image", i, (rows, cols), kw, "lines", len(L), "vs", len(ref["lines"]), "| equals the correctly rounded restatement:", eq, flush=True)
            if not eq:
                np.save(os.path.join(ROOT, "gpurun_out", "campaign_batch_bad_%d_%d.npy" % (gi, i)), batch[i])
print("batch campaign: %d groups x %d images = %d images, %d mismatches, %d seeds evaluated by helpers, %.0f s" % (groups, per, tot, bad, helped, time.time() - t0))
