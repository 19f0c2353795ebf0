#!/usr/bin/env python3
"""Developer probe: calls of 1 .. 64 images with the help across workgroups at its default and off (is the helper pool / the help for
small calls still worth it after the certified sets?).   tools/help_small_probe.py"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, torch
lsd = importlib.import_module("linesegmentdetector-slam_amd")
maps = bench.load_maps()
size = 2048
sets = {"first": list(range(64)), "heavy": [27, 396, 33, 220, 1, 16, 187, 45] + list(range(100, 156))}
for sname, ids in sets.items():
    full = torch.from_numpy(np.stack([bench.make_image(maps, i, size) for i in ids])).cuda()
    out = torch.zeros((64, 1024, 10), dtype=torch.int64, device="cuda"); cnt = torch.zeros(64, dtype=torch.int32, device="cuda")
    for n in (1, 2, 4, 8, 16, 32, 64):
        row = []
        for help_ in (-1, 0):
            c = lsd.Context(0); c.set_region_help(help_); c.reserve(n, size, size)
            ts = []
            for rep in range(5):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                c.enqueue_device(full.data_ptr(), n, size, size, out.data_ptr(), 1024, cnt.data_ptr()); torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            row.append((float(np.median(ts[1:])), int(cnt[:n].sum())))
            c.close()
        print("%-5s n %2d: help default %.1f ms, help off %.1f ms (lines %d / %d)" % (sname, n, row[0][0], row[1][0], row[0][1], row[1][1]), flush=True)
# the reference's own maps, one per call (host entry point)
for name in bench.REAL_MAPS:
    row = []
    for help_ in (-1, 0):
        c = lsd.Context(0); c.set_region_help(help_)
        ts = []
        for rep in range(12):
            t0 = time.perf_counter(); r = c.run(maps[name].copy()); ts.append((time.perf_counter() - t0) * 1e3)
        row.append(float(np.median(ts[2:]))); c.close()
    print("%-8s one host call: help default %.2f ms, help off %.2f ms" % (name, row[0], row[1]), flush=True)
