#!/usr/bin/env python3
"""Developer probe: per-image region-stage cycles for the whole bench batch; prints the distribution and the slowest images."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, torch
lsd = importlib.import_module("linesegmentdetector-slam_amd")
maps = bench.load_maps()
n, size = int(sys.argv[1]) if len(sys.argv) > 1 else 512, 2048
ctx = lsd.Context(0)
d = torch.from_numpy(bench.make_batch(maps, n, size)).cuda()
lines = torch.zeros((n, 1024, 10), dtype=torch.int64, device="cuda"); counts = torch.zeros(n, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for rep in range(2):
    ctx.enqueue_device(d.data_ptr(), n, size, size, lines.data_ptr(), 1024, counts.data_ptr(), stream=s); torch.cuda.synchronize()
print({k: round(v, 2) for k, v in ctx.timings().items()})
wh = lsd.scaled_size(size, size)
st = [ctx.fetch(i, lsd.DBG_STATS, wh) for i in range(n)]
tot = np.array([x["cycles_total"] for x in st]) / 1e6
print("Mcycles total: mean %.0f median %.0f p90 %.0f max %.0f" % (tot.mean(), np.median(tot), np.percentile(tot, 90), tot.max()))
for i in np.argsort(-tot)[:6]:
    x = st[i]
    print(i, "src", bench.SOURCES[i % 4], "flip", (i // 4) % 4, {k: (v // 1000000 if k.startswith("cycles") else v) for k, v in x.items()})
print("per source: n, mean/max Mcycles total, then mean wave-Mcycles of wait / eval / idle / commit")
for si, name in enumerate(bench.SOURCES):
    idx = [i for i in range(n) if i % 4 == si]
    f = lambda key: np.mean([st[i][key] for i in idx]) / 1e6
    print("  %-10s %3d  %5.0f %5.0f   wait %5.0f eval %5.0f (grow %5.0f sums %4.0f rect %4.0f nfa %4.0f refine %4.0f) idle %5.0f commit %4.0f  seeds %6.0f grows %6.0f lines %s" % (
        name, len(idx), tot[idx].mean(), tot[idx].max(), f("cycles_wait"), f("cycles_eval"), f("cycles_grow"), f("cycles_sums"), f("cycles_rect"), f("cycles_nfa"), f("cycles_refine"), f("cycles_small"), f("cycles_commit"),
        np.mean([st[i]["seeds"] for i in idx]), np.mean([st[i]["grow_calls"] for i in idx]), ""))
