#!/usr/bin/env python3
"""Developer tool: randomized parity campaign for createMapCache on the device (both launch forms) against the oracle."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import oracle
lsd = importlib.import_module("linesegmentdetector-slam_amd")
oracle.build()
ctx = lsd.Context(0)
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = 0; t0 = time.time()
for i in range(n_img):
    rng = np.random.default_rng(50_000 + i)
    rows, cols = int(rng.integers(1, 500)), int(rng.integers(1, 700))
    m = np.zeros((rows, cols), np.uint8)
    m[rng.random((rows, cols)) < rng.uniform(0, 0.4)] = 255
    m[rng.random((rows, cols)) < rng.choice([0.0, 0.0005, 0.01, 0.2, 1.0])] = 1
    res = float(rng.choice([0.025, 0.05, 0.1, 0.3, 1.0, 2.5]))
    zmax = float(rng.choice([1.0, 2.0, 0.5]))
    ref = oracle.map_cache(m.copy(), res, zmax)
    got = ctx.map_cache(m, res, zmax)                                   # one map: spread over workgroups
    nb = int(rng.choice([3, 140]))                                      # small batch: spread; > 128 maps: one workgroup per map
    d = torch.from_numpy(np.broadcast_to(m, (nb, rows, cols)).copy()).cuda()
    out = torch.zeros((nb, rows, cols), dtype=torch.float64, device="cuda")
    ctx.enqueue_map_cache_device(d.data_ptr(), nb, cols, rows, res, zmax, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    if not (np.array_equal(got, ref) and np.array_equal(o[0], ref) and np.array_equal(o[-1], ref)):
        bad += 1; print("MISMATCH", i, (rows, cols), res, zmax, nb, flush=True)
print("mapcache campaign: %d maps, %d mismatches, %.0f s" % (n_img, bad, time.time() - t0))
