#!/usr/bin/env python3
"""Developer probe: first seed whose decision differs from the oracle."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle
lsd = importlib.import_module("linesegmentdetector-slam_amd")
z = np.load(os.path.join(ROOT, "tests/golden/maps.npz"))
name = sys.argv[1]
img = z[name]
ref = oracle.lsd(img.copy(), debug=True); d = ref["dbg"]
ctx = lsd.Context(0); ctx.set_trace(True)
lines, im = ctx.run(img.copy())
w, h = d["w"], d["h"]
seeds = ctx.fetch(0, lsd.DBG_SEEDS, (w, h)); rs = d["seeds"]
print("nseed", len(seeds), len(rs), "lines", len(lines), len(ref["lines"]))
k = min(len(seeds), len(rs))
for i in range(k):
    a, b = seeds[i], rs[i]
    if any(a[f] != b[f] for f in ("order_idx", "num", "outcome", "final_num")) or abs(a["logNFA"] - b["logNFA"]) > 1e-9 * max(1, abs(b["logNFA"])):
        print("seed", i, "gpu", a, "ref", b)
        if any(a[f] != b[f] for f in ("order_idx", "num", "outcome", "final_num")):
            break
used = (ctx.fetch(0, lsd.DBG_STATE, (w, h)) & 3).astype(np.uint8)
ys, xs = np.nonzero(used != d["used"])
print("used diffs", len(ys), list(zip(ys[:10], xs[:10])))
dg = ctx.fetch(0, lsd.DBG_DEG, (w, h))
du = np.abs(dg.view(np.int64) - d["deg"].view(np.int64))
print("deg ulp diffs", int((du > 0).sum()), int(du.max()))
