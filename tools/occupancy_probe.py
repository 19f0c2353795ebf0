#!/usr/bin/env python3
"""Developer probe: how full the GPU is in the timed configuration (eight steps in flight, 4-wave region stage, help off).
Every image's workgroup records when it ran (s_memrealtime at its start and end, its XCC); the probe reads those records of each
step before the step's slot is used again and prints the number of resident region-stage workgroups over a steady window
(768 = three per CU), per XCC, next to the event times of the step's kernels.
   tools/occupancy_probe.py [depth [steps]]"""
import importlib, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
lsd = importlib.import_module("linesegmentdetector-slam_amd")
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = int(sys.argv[2]) if len(sys.argv) > 2 else 48
maps = bench.load_maps(); n, size = 512, 2048
d = torch.from_numpy(bench.make_batch(maps, n, size)).cuda()
wh = lsd.scaled_size(size, size)
ctxs = [lsd.Context(0) for _ in range(depth)]
for c in ctxs:
    c.set_region_waves(4); c.set_region_help(0); c.reserve(n, size, size)
streams = [torch.cuda.Stream() for _ in range(depth)]
outs = [(torch.zeros((n, 1024, 10), dtype=torch.int64, device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda"),
         torch.zeros((n, size, size), dtype=torch.uint8, device="cuda")) for _ in range(depth)]
def step(i):
    j = i % depth
    l, c, im = outs[j]
    ctxs[j].enqueue_device(d.data_ptr(), n, size, size, l.data_ptr(), 1024, c.data_ptr(), d_line_ims=im.data_ptr(), stream=streams[j].cuda_stream)
lo, hi = 2 * depth, K - depth          # steps whose records are read
rec = {}
tms = {}
t0 = time.perf_counter()
for i in range(K + depth):
    j = i % depth
    prev = i - depth
    if lo <= prev < hi:
        st = ctxs[j].fetch_stats_block(n)                 # slots 46 / 47: s_memrealtime (100 MHz) at the image's start / end, 45: its XCC, 8: clocks
        rec[prev] = st[:, [46, 47, 45, 8]].astype(float)
        tms[prev] = ctxs[j].timings()
    if i < K: step(i)
torch.cuda.synchronize()
print("depth %d, %d steps: %.1f ms per step (with the reads)" % (depth, K, (time.perf_counter() - t0) * 1e3 / K))
steps = sorted(rec)
A = np.concatenate([rec[s] for s in steps]); b0, e0 = A[:, 0] / 1e5, A[:, 1] / 1e5
med = {s: np.median(rec[s][:, 0]) / 1e5 for s in steps}
T0, T1 = med[steps[depth]], med[steps[-depth]]      # a window every instant of which is covered by the steps read
print("window %.1f ms = %d steps -> %.2f ms per step" % (T1 - T0, len(steps) - 2 * depth, (T1 - T0) / (len(steps) - 2 * depth)))
ts = np.arange(T0, T1, 0.05)
res = np.zeros_like(ts); rx = np.zeros((8, len(ts)))
for (b, e, x) in zip(b0, e0, A[:, 2].astype(int)):
    i0, i1 = np.searchsorted(ts, [b, e])
    res[i0:i1] += 1; rx[x & 7, i0:i1] += 1
print("resident region workgroups: mean %.0f of 768 (p10 %.0f p50 %.0f p90 %.0f max %.0f); per XCC mean: %s" % (
    res.mean(), *np.percentile(res, [10, 50, 90]), res.max(), " ".join("%.0f" % v for v in rx.mean(1))))
print("share of the window with < 384 resident: %.2f, < 576: %.2f, >= 700: %.2f" % ((res < 384).mean(), (res < 576).mean(), (res >= 700).mean()))
print("workgroup time per image: mean %.1f ms (%.0f Mcycles), max %.1f ms" % ((e0 - b0).mean(), A[:, 3].mean() / 1e6, (e0 - b0).max()))
for s in steps[depth:depth + 10]:
    r = rec[s]; t = tms[s]; b = r[:, 0] / 1e5; e = r[:, 1] / 1e5
    print("step %d: region starts %.1f..%.1f (p50 %.1f, p90 %.1f), ends p50 %.1f p90 %.1f last %.1f | events: gauss %.1f gradient %.1f sort %.1f region %.1f lines %.1f" % (
        s, b.min() - T0, b.max() - T0, np.median(b) - T0, np.percentile(b, 90) - T0, np.median(e) - T0, np.percentile(e, 90) - T0, e.max() - T0,
        t["gauss"], t["gradient"], t["sort"], t["region"], t["lines"]))
# coarse timeline: resident workgroups every 5 ms over the first 200 ms of the window
print("resident every 5 ms:", " ".join("%.0f" % res[k:k + 100].mean() for k in range(0, min(len(res), 4000), 100)))
