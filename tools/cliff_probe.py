#!/usr/bin/env python3
"""Developer probe: the region stage's cycles per scaled pixel (lsd_last_region_cycles) of the giant-region images against occupancy
maps -- the number INTEGRATION.md gives callers to recognise an image the region stage is slow on.   tools/cliff_probe.py"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, torch  # noqa: F401
lsd = importlib.import_module("linesegmentdetector-slam_amd")
from test_parity_gpu import _sawtooth
from oracle import oracle
ctx = lsd.Context(0)
maps = bench.load_maps()
todo = [(k, _sawtooth(k)) for k in ("saw_x", "saw_diag", "saw_tall", "saw_noise")] + [(k, maps[k]) for k in bench.REAL_MAPS] + \
       [("bench image %d" % i, bench.make_image(maps, i, 2048)) for i in (0, 1, 27, 187)]
for name, img in todo:
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); ctx.run(img.copy()); ts.append(time.perf_counter() - t0)
    cyc = int(ctx.last_region_cycles(1)[0])
    w, h = lsd.scaled_size(img.shape[1], img.shape[0])
    t0 = time.perf_counter(); oracle.lsd(img.copy()); tc = time.perf_counter() - t0
    print("%-16s %5d x %-5d  region %7.1f M cycles = %6.0f per scaled pixel; host call %6.1f ms; one host thread of the restatement %7.1f ms" % (
        name, img.shape[1], img.shape[0], cyc / 1e6, cyc / (w * h), min(ts) * 1e3, tc * 1e3), flush=True)
