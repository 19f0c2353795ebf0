#!/usr/bin/env python3
"""Developer tool: the whole 512-image bench batch through the device entry point, every image compared with the oracle
(line count, lineIm, line records)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from oracle import oracle
lsd = importlib.import_module("linesegmentdetector-slam_amd")
oracle.build()
maps = bench.load_maps()
n, size = int(sys.argv[1]) if len(sys.argv) > 1 else 512, 2048
host = bench.make_batch(maps, n, size)
ctx = lsd.Context(0)
d = torch.from_numpy(host).cuda()
lines = torch.zeros((n, 1024, 10), dtype=torch.int64, device="cuda"); counts = torch.zeros(n, dtype=torch.int32, device="cuda")
ims = torch.zeros((n, size, size), dtype=torch.uint8, device="cuda")
ctx.enqueue_device(d.data_ptr(), n, size, size, lines.data_ptr(), 1024, counts.data_ptr(), d_line_ims=ims.data_ptr(),
                   stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
cnt = counts.cpu().numpy(); L = lines.cpu().numpy().view(np.uint8).reshape(n, 1024, 80)
bad = 0; t0 = time.time()
for i in range(n):
    ref = oracle.lsd(host[i].copy())
    rl = ref["lines"]
    ok = cnt[i] == len(rl) and np.array_equal(ims[i].cpu().numpy(), ref["lineIm"])
    if ok and len(rl):
        gl = L[i, :cnt[i]].copy().view(lsd.LINE_DTYPE).reshape(-1)
        ok = all(np.abs(gl[f] - rl[f]).max() < 1e-6 for f in ("x1", "y1", "x2", "y2")) and np.array_equal(gl["orient"], rl["orient"])
    if not ok:
        bad += 1; print("MISMATCH image", i, "lines", cnt[i], "vs", len(rl), flush=True)
print("bench batch: %d images, %d mismatches, total lines %d, oracle %.0f s" % (n, bad, cnt.sum(), time.time() - t0))
