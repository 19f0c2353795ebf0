#!/usr/bin/env python3
"""Developer probe: the same batch through the device entry point many times -- line counts and records must be byte-identical
every time (speculation, commit order and waves finishing in different orders must not show).   tools/determinism_probe.py [n] [reps]"""
import importlib, os, sys, hashlib, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
lsd = importlib.import_module("linesegmentdetector-slam_amd")
maps = bench.load_maps()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
size = 2048
ctx = lsd.Context(0)
d0 = bench.make_batch(maps, n, size)
lines = torch.zeros((n, 1024, 10), dtype=torch.int64, device="cuda"); counts = torch.zeros(n, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
ref = None
for rep in range(reps):
    d = torch.from_numpy(d0.copy()).cuda()
    lines.zero_()
    ctx.enqueue_device(d.data_ptr(), n, size, size, lines.data_ptr(), 1024, counts.data_ptr(), stream=s); torch.cuda.synchronize()
    c = counts.cpu().numpy().copy()
    L = lines.cpu().numpy()
    h = hashlib.sha1(np.concatenate([L[i, :c[i]].ravel() for i in range(n)]).tobytes()).hexdigest()
    if ref is None: ref = (c, h)
    print(rep, int(c.sum()), h[:12], "same" if (np.array_equal(c, ref[0]) and h == ref[1]) else "DIFFERENT", flush=True)
