#!/usr/bin/env python3
"""Developer probe: device memory a context reserves for a batch of 2048 x 2048 maps, per region-stage variant.   tools/workspace_size.py"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
lsd = importlib.import_module("linesegmentdetector-slam_amd")
torch.zeros(1, device="cuda")
for waves in (4, 8):
    for n in (512, 64):
        f0 = torch.cuda.mem_get_info()[0]
        c = lsd.Context(0); c.set_region_waves(waves); c.reserve(n, 2048, 2048)
        f1 = torch.cuda.mem_get_info()[0]
        print("waves %d, %d images: workspace %.2f GB = %.1f MB per image" % (waves, n, (f0 - f1) / 1e9, (f0 - f1) / 1e6 / n))
        c.close()
