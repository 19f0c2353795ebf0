import importlib, os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from oracle import oracle
lsd = importlib.import_module("linesegmentdetector-slam_amd")
z = np.load("tests/golden/maps.npz"); name = sys.argv[1]; img = z[name]
d = oracle.lsd(img.copy(), debug=True)["dbg"]
ctx = lsd.Context(0); ctx.set_trace(True)
ref = oracle.lsd(img.copy())
for rep in range(8):
    lines, im = ctx.run(img.copy())
    used = (ctx.fetch(0, lsd.DBG_STATE, (d['w'], d['h'])) & 3).astype(np.uint8)
    print(rep, 'lines', len(lines), len(ref['lines']), 'used_ok', bool(np.array_equal(used, d['used'])), 'im_ok', bool(np.array_equal(im, ref['lineIm'])), end=' ')
    seeds = ctx.fetch(0, lsd.DBG_SEEDS, (d["w"], d["h"])); rs = d["seeds"]
    g = {int(s["order_idx"]): s for s in seeds}; r = {int(s["order_idx"]): s for s in rs}
    miss = [k for k in r if k not in g]; extra = [k for k in g if k not in r]
    dup = len(seeds) - len(g)
    print(rep, "gpu", len(seeds), "ref", len(rs), "missing", [(k, int(r[k]["outcome"])) for k in miss[:8]], "extra", extra[:5], "dups", dup,
          "order ok", all(a <= b for a, b in zip(seeds["order_idx"][:-1], seeds["order_idx"][1:])))
