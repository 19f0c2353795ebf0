#!/usr/bin/env python3
"""Developer tool: re-runs one image of tools/campaign.py (by index) in every region-stage mode and prints the first seed whose
record differs from the oracle's trace."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: F401
from oracle import oracle
lsd = importlib.import_module("linesegmentdetector-slam_amd")
oracle.build()
import types
src = open(os.path.join(ROOT, "tools", "campaign.py")).read().split("bad = 0")[0].replace("n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 200", "n_img = 0")
ns = {"__file__": os.path.join(ROOT, "tools", "campaign.py")}
exec(compile(src, "campaign_head", "exec"), ns)
synth, ctx = ns["synth"], ns["ctx"]
i = int(sys.argv[1])
rng = np.random.default_rng(10_000 + i)
img = synth(rng)
kw = {}
if rng.random() < 0.3:
    kw = dict(sca=0.3, sig=float(rng.choice([0.6, 0.8])), angThre=float(rng.choice([22.5, 20.0, 30.0])),
              denThre=float(rng.choice([0.7, 0.6])), pseBin=int(rng.choice([1024, 512, 256])))
ref = oracle.lsd(img.copy(), debug=True, **kw)
d = ref["dbg"]; rs = d["seeds"]
ctx.set_trace(True)
np.save(os.path.join(ROOT, "gpurun_out", "campaign_img_%d.npy" % i), img)
for mode in (4, 8, 4, 8):
    ctx.set_region_waves(mode)
    lines, im = ctx.run(img.copy(), lsd.make_params(**kw) if kw else None)
    seeds = ctx.fetch(0, lsd.DBG_SEEDS, (d["w"], d["h"]))
    used = (ctx.fetch(0, lsd.DBG_STATE, (d["w"], d["h"])) & 3).astype(np.uint8)
    first = None
    for a, b in zip(seeds, rs):
        if (a["order_idx"], a["num"], a["outcome"], a["final_num"]) != (b["order_idx"], b["num"], b["outcome"], b["final_num"]):
            first = (a, b); break
    print("mode", mode, "lines", len(lines), "vs", len(ref["lines"]), "usedMap diff", int((used != d["used"]).sum()), "trace", len(seeds), len(rs))
    if first:
        print("   first differing seed: gpu", first[0], "\n                          ref", first[1])
