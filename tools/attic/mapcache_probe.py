#!/usr/bin/env python3
"""Developer probe: createMapCache on the device (k_mapcache) vs the CPU restatement, 2048^2 maps."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from oracle import oracle
lsd = importlib.import_module("linesegmentdetector-slam_amd")
oracle.build()
maps = bench.load_maps()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
host = bench.make_batch(maps, n, 2048)
ctx = lsd.Context(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
d = torch.from_numpy(host).cuda()
out = torch.zeros((n, 2048, 2048), dtype=torch.float64, device="cuda")
def run(k):
    ctx.enqueue_map_cache_device(d.data_ptr(), k, 2048, 2048, 0.025, 1.0, out.data_ptr(), stream=st.cuda_stream)
for k in (1, n):
    run(k); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(k); run(k); e1.record(); torch.cuda.synchronize()
    print("k_mapcache %d x 2048^2: %.2f ms per launch (%.1f Mpix/s)" % (k, e0.elapsed_time(e1) / 2, k * 2048 * 2048 / (e0.elapsed_time(e1) / 2) / 1e3))
t0 = time.perf_counter(); ref = oracle.map_cache(host[0].copy(), 0.025); cpu = time.perf_counter() - t0
print("CPU restatement: %.1f ms per map; equal: %s" % (cpu * 1e3, np.array_equal(out[0].cpu().numpy(), ref)))
