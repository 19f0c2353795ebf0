#!/usr/bin/env python3
"""Streams 2 GiB with 8-B-per-lane stores then loads (run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lsd = importlib.import_module("linesegmentdetector-slam_amd")
ctx = lsd.Context(0)
ctx._chk(ctx.L.lsd_debug_calibrate(ctx.h, 2 << 30))
print("calibrated over", 2 << 30, "bytes")
