#!/usr/bin/env python3
"""Developer probe: the scan-to-map matching batch (k_match) on the device vs the CPU restatement, one synthetic frame."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import oracle
from matching_case import build_case
lsd = importlib.import_module("linesegmentdetector-slam_amd")
oracle.build()
z = np.load(os.path.join(ROOT, "tests", "golden", "maps.npz"))
m = np.ascontiguousarray(np.tile(z["aisle1"], (4, 2))[:2048, :2048])           # the 2048^2 tile: 238 map lines
case = build_case(m, 0.025, oracle, theta_deg=17.0, centre=(700.0, 600.0), half=300, max_points=1000)
# every scan line against every map line (the reference prunes by length; this is the upper bound of a frame)
pairs = np.array([(a, b) for b in range(len(case["scan_lines"])) for a in range(len(case["map_lines"]))], np.int32)
ctx = lsd.Context(0)
rows, cols = m.shape
st = torch.cuda.Stream(); torch.cuda.set_stream(st)      # a real stream: handle 0 would make the library use its own
s = st.cuda_stream
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).cuda()
d_mc, d_ml, d_sl, d_pts, d_pr = dev(case["map_cache"]), dev(case["map_lines"]), dev(case["scan_lines"]), dev(case["pts"]), dev(pairs)
d_out = torch.zeros((len(pairs) * 4, 4), dtype=torch.float64, device="cuda")
P = lsd.lsd_position
def run():
    ctx._chk(ctx.L.lsd_enqueue_scan_to_map_match_device(ctx.h, d_mc.data_ptr(), cols, rows, d_ml.data_ptr(), d_sl.data_ptr(), d_pts.data_ptr(),
             len(case["pts"]), P(*case["lidar"]), P(-1.0, -1.0, 0.0), d_pr.data_ptr(), len(pairs), 1.0, 60.0, d_out.data_ptr(), s))
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
t0 = time.perf_counter()
want = oracle.scan_to_map_match(case["map_cache"], case["map_lines"], case["scan_lines"], case["pts"], case["lidar"], (-1.0, -1.0, 0.0), pairs)
cpu = time.perf_counter() - t0
g = d_out.cpu().numpy(); w = want.reshape(-1, 4); fin = np.isfinite(w[:, 3])
print("pairs %d candidates %d points %d: device %.3f ms (%.1f M point-evals/s), CPU restatement 1 thread %.1f ms, max |diff| %.2e, inf pattern equal %s"
      % (len(pairs), 4 * len(pairs), len(case["pts"]), ms, 4 * len(pairs) * len(case["pts"]) / ms / 1e3, cpu * 1e3,
         np.abs(g[fin] - w[fin]).max(), np.array_equal(np.isinf(g[:, 3]), ~fin)))
