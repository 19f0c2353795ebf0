import importlib, os, sys, numpy as np
sys.path.insert(0, os.getcwd())
lsd = importlib.import_module("linesegmentdetector-slam_amd")
z = np.load("tests/golden/maps.npz"); ctx = lsd.Context(0)
def u4(v): return [np.int16((v >> s) & 0xffff) for s in (48, 32, 16, 0)]
for name in ("aisle1", "f3key"):
    img = z[name]; ctx.run(img.copy())
    st = ctx.fetch(0, lsd.DBG_STATS, lsd.scaled_size(img.shape[1], img.shape[0]))
    v = st["_r31"]
    print(name, "missed:", st["_r23"], "k=%d n_spec=%d n_turn=%d" % (v >> 32, (v >> 16) & 0xffff, v & 0xffff),
          "snap,now", st["pt_pick"] >> 32, st["pt_pick"] & 0xffffffff, "box", u4(st["pt_reads"]), "ring[snap]", u4(st["pt_classify"]),
          "seed", st["pt_chain"] >> 32, st["pt_chain"] & 0xffffffff)
