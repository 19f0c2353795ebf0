#!/usr/bin/env python3
"""Developer experiment: which cheap per-image features predict the region stage's cost?  Runs the bench batch (helpers off), takes the
per-image cycles, computes candidate features from the angle map and the usedMap after the gradient pass (LSD_STOP... not needed:
the angles do not change), and reports correlations and a cross-validated linear fit."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["LSD_REGION_HELP"] = "0"
import bench, torch
lsd = importlib.import_module("linesegmentdetector-slam_amd")
maps = bench.load_maps(); n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = lsd.Context(0)
d = torch.from_numpy(bench.make_batch(maps, n, 2048)).cuda()
lines = torch.zeros((n, 1024, 10), dtype=torch.int64, device="cuda"); counts = torch.zeros(n, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for rep in range(2):
    ctx.enqueue_device(d.data_ptr(), n, 2048, 2048, lines.data_ptr(), 1024, counts.data_ptr(), stream=s); torch.cuda.synchronize()
wh = lsd.scaled_size(2048, 2048)
cyc = np.array([ctx.fetch(i, lsd.DBG_STATS, wh)["cycles_total"] for i in range(n)], float)
tol = 22.5 / 180 * np.pi
F = []
for i in range(n):
    deg = ctx.fetch(i, lsd.DBG_DEG, wh); mag = ctx.fetch(i, lsd.DBG_MAG, wh)
    nb = ctx.fetch(i, lsd.DBG_NB, wh)
    thr = 2.0 / np.sin(tol)
    free = mag >= thr                                              # not banned by the gradient threshold
    cnt = np.zeros(deg.shape, np.int32)
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            if dx == 0 and dy == 0: continue
            a = np.roll(np.roll(deg, dy, 0), dx, 1); f2 = np.roll(np.roll(free, dy, 0), dx, 1)
            df = np.abs(a - deg); df = np.where(df > np.pi * 1.5, np.abs(df - 2 * np.pi), df)
            cnt += (free & f2 & (df < tol)).astype(np.int32)
    F.append([nb, free.sum(), cnt[free].sum(), (cnt[free] >= 2).sum(), (cnt[free] >= 4).sum(), (cnt[free] >= 6).sum(), (mag[free] > 4 * thr).sum()])
F = np.array(F, float)
names = ["nb", "free px", "sum aligned nbrs", ">=2 aligned", ">=4 aligned", ">=6 aligned", "mag > 4 thr"]
for j, nm in enumerate(names): print("corr(%s, cycles) = %.2f" % (nm, np.corrcoef(F[:, j], cyc)[0, 1]))
X = np.c_[F, np.ones(n)]
idx = np.arange(n); pred = np.zeros(n)
for fold in range(4):
    te = idx % 4 == fold
    coef, *_ = np.linalg.lstsq(X[~te], cyc[~te], rcond=None)
    pred[te] = X[te] @ coef
print("4-fold linear fit on all features: corr %.2f; top-64 overlap %d of 64; bottom-256 contains %d of the 64 costliest" % (
    np.corrcoef(pred, cyc)[0, 1], len(set(np.argsort(-pred)[:64]) & set(np.argsort(-cyc)[:64])), len(set(np.argsort(pred)[:256]) & set(np.argsort(-cyc)[:64]))))
print("by nb alone: bottom-256 contains %d of the 64 costliest" % len(set(np.argsort(F[:, 0])[:256]) & set(np.argsort(-cyc)[:64])))
np.save("gpurun_out/cost_features.npy", np.c_[F, cyc])
