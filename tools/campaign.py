#!/usr/bin/env python3
"""Developer tool: randomized parity campaign -- many synthetic maps (walls, noise, random sizes, random parameters) through the
HIP path and the oracle; reports every image whose usedMap, line count, lineIm or line records differ."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (before the HIP library)
from oracle import oracle
lsd = importlib.import_module("linesegmentdetector-slam_amd")
oracle.build()
ctx = lsd.Context(0)
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 200
only = [int(x) for x in sys.argv[2:]]                      # optional: just these image numbers


def synth(rng):
    big = os.environ.get("CAMPAIGN_BIG")                                  # larger maps (spill paths, long lists): CAMPAIGN_BIG=1
    rows, cols = (int(rng.integers(1500, 3500)), int(rng.integers(1500, 3500))) if big else (int(rng.integers(60, 900)), int(rng.integers(60, 1200)))
    m = np.zeros((rows, cols), np.uint8)
    m[rng.random((rows, cols)) < rng.uniform(0.0, 0.5)] = 255
    for _ in range(int(rng.integers(3, 160 if os.environ.get("CAMPAIGN_BIG") else 40))):
        x0, y0 = rng.integers(2, cols - 2), rng.integers(2, rows - 2)
        L = int(rng.integers(10, 2500 if big else 400)); a = rng.choice([0, np.pi / 2, np.pi / 4, rng.uniform(0, np.pi)])
        t = np.arange(L)
        xs = np.clip((x0 + t * np.cos(a)).astype(int), 0, cols - 1); ys = np.clip((y0 + t * np.sin(a)).astype(int), 0, rows - 1)
        m[ys, xs] = 1
        if rng.random() < 0.3:                                        # thick wall
            m[np.clip(ys + 1, 0, rows - 1), xs] = 1
    if rng.random() < 0.3:                                            # salt noise of occupied cells
        m[rng.random((rows, cols)) < 0.01] = 1
    return m


bad = 0
nfa_abs, nfa_gap = float('inf'), float('inf')      # closest any NFA decision of the campaign came to a tie
t0 = time.time()
for i in (only or range(n_img)):
    rng = np.random.default_rng(10_000 + i)
    img = synth(rng)
    kw = {}
    if rng.random() < 0.3:
        kw = dict(sca=0.3, sig=float(rng.choice([0.6, 0.8])), angThre=float(rng.choice([22.5, 20.0, 30.0])),
                  denThre=float(rng.choice([0.7, 0.6])), pseBin=int(rng.choice([1024, 512, 256])))
    ctx.set_region_waves(int(rng.choice([0, 4, 8])))
    ref = oracle.lsd(img.copy(), debug=True, **kw)
    d = ref["dbg"]
    lines, im = ctx.run(img.copy(), lsd.make_params(**kw) if kw else None)
    used = (ctx.fetch(0, lsd.DBG_STATE, (d["w"], d["h"])) & 3).astype(np.uint8)
    st_ = ctx.fetch(0, lsd.DBG_STATS, (d["w"], d["h"]))
    nfa_abs, nfa_gap = min(nfa_abs, st_["nfa_min_abs"]), min(nfa_gap, st_["nfa_min_gap"])
    ok = len(lines) == len(ref["lines"]) and np.array_equal(used, d["used"]) and np.array_equal(im, ref["lineIm"])
    if ok and len(lines):
        ok = all(np.abs(lines[f] - ref["lines"][f]).max() < 1e-6 for f in ("x1", "y1", "x2", "y2")) and np.array_equal(lines["orient"], ref["lines"]["orient"])
    if not ok:
        bad += 1
        # the diagnostic variant with correctly rounded sin/cos/atan2 (oracle/cr_shim.cpp): a libm tie if the HIP path equals it
        rc = oracle.lsd(img.copy(), debug=True, _lib=oracle.lib_cr(), **kw)
        eq = (len(lines) == len(rc["lines"]) and np.array_equal(used, rc["dbg"]["used"]) and np.array_equal(im, rc["lineIm"]) and
              (len(lines) == 0 or all(np.abs(lines[f] - rc["lines"][f]).max() < 1e-6 for f in ("x1", "y1", "x2", "y2"))))
        print("MISMATCH image", i, img.shape, kw, "lines", len(lines), "vs", len(ref["lines"]), "usedMap diff", int((used != d["used"]).sum()),
              "| equals the correctly rounded restatement:", eq, flush=True)
        if not eq:
            np.save(os.path.join(ROOT, "gpurun_out", "campaign_bad_%d.npy" % i), img)
print("campaign: %d images, %d mismatches, %.0f s; smallest |logNFA| compared with 0: %.3g, smallest non-zero gap between compared NFA values: %.3g" % (n_img, bad, time.time() - t0, nfa_abs, nfa_gap))
