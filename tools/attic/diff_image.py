#!/usr/bin/env python3
"""Developer probe: one bench image through the HIP path and the oracle; prints where they differ (lines, seed trace)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from oracle import oracle
lsd = importlib.import_module("linesegmentdetector-slam_amd")
maps = bench.load_maps(); ctx = lsd.Context(0)
i = int(sys.argv[1]); waves = int(sys.argv[2]) if len(sys.argv) > 2 else 0
img = bench.make_image(maps, i, 2048)
ref = oracle.lsd(img.copy(), debug=True)
d = ref["dbg"]
ctx.set_region_waves(waves); ctx.set_trace(True)
lines, im = ctx.run(img.copy())
seeds = ctx.fetch(0, lsd.DBG_SEEDS, (d["w"], d["h"]))
print("lines", len(lines), len(ref["lines"]), "lineIm equal", np.array_equal(im, ref["lineIm"]))
for j, (a, b) in enumerate(zip(lines, ref["lines"])):
    e = max(abs(a[f] - b[f]) for f in ("x1", "y1", "x2", "y2"))
    if e > 1e-9: print("line", j, "maxdiff", e, [a[f] for f in ("x1", "y1", "x2", "y2")], [b[f] for f in ("x1", "y1", "x2", "y2")])
rs = d["seeds"]
print("seeds", len(seeds), len(rs))
m = min(len(seeds), len(rs))
for f in ("order_idx", "num", "outcome", "final_num"):
    bad = np.nonzero(seeds[f][:m] != rs[f][:m])[0]
    if len(bad): print(f, "first diffs at", bad[:5], seeds[bad[:3]], rs[bad[:3]]); break
ev = rs["outcome"][:m] >= 2
bad = np.nonzero(ev & (seeds["logNFA"][:m] != rs["logNFA"][:m]))[0]
print("logNFA diffs", len(bad), [(int(b), float(seeds["logNFA"][b]), float(rs["logNFA"][b])) for b in bad[:5]])
recs = ctx.fetch_recs(0, len(lines))
rr = d["recs"]
bad = np.nonzero(np.abs(recs - rr[:len(recs)]).max(1) > 0)[0]
print("rec diffs (bitwise)", len(bad), bad[:10])
for b in bad[:3]: print(recs[b], rr[b], recs[b] - rr[b])
