#!/usr/bin/env python3
"""Developer probe: throughput mode of the sharded job projected on one GPU -- every shard of the 512-image batch with D steps in flight
(4-wave region stage, help off; D = 8 x N, at most 32), contiguous shards against interleaved ones (image i on rank i % N).
   tools/shard_probe.py N[,N...] [depth]      several N: one after the other in this process
   ONLY=contiguous|interleaved                 one layout only
   HISTORY=1                                   one 1-image launch on a side stream BEFORE the slots' contexts are created: reproduces the
                                               1.3-1.4 x slower 32-steps-in-flight runs seen inside bench.py's process (DESIGN_NOTES.md)"""
import importlib, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
lsd = importlib.import_module("linesegmentdetector-slam_amd")
Ns = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [8]
depth0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
maps = bench.load_maps(); n, size = 512, 2048
d = torch.from_numpy(bench.make_batch(maps, n, size)).cuda()
only = os.environ.get("ONLY", "")
def outputs(m):
    return (torch.zeros((m, 1024, 10), dtype=torch.int64, device="cuda"), torch.zeros(m, dtype=torch.int32, device="cuda"),
            torch.zeros((m, size, size), dtype=torch.uint8, device="cuda"))
if os.environ.get("HISTORY"):
    c0 = lsd.Context(0); c0.set_region_waves(4); c0.set_region_help(0)
    o = outputs(1); side = torch.cuda.Stream()
    c0.enqueue_device(d.data_ptr(), 1, size, size, o[0].data_ptr(), 1024, o[1].data_ptr(), d_line_ims=o[2].data_ptr(), stream=side.cuda_stream)
    torch.cuda.synchronize()
    if os.environ["HISTORY"] == "close": c0.close()          # (makes no difference)
for N in Ns:
    depth = depth0 or min(8 * N, 32)
    m = n // N
    slots = []
    for j in range(depth):
        c = lsd.Context(0); c.set_region_waves(4); c.set_region_help(0); c.reserve(m, size, size)
        slots.append((c, torch.cuda.Stream()) + outputs(m))
    def run(shard):
        def go(i):
            c, st, l, cn, im = slots[i % depth]
            c.enqueue_device(shard.data_ptr(), m, size, size, l.data_ptr(), 1024, cn.data_ptr(), d_line_ims=im.data_ptr(), stream=st.cuda_stream)
        for i in range(depth): go(i)
        torch.cuda.synchronize()
        steps = 6 * depth
        ev0 = torch.cuda.Event(enable_timing=True); evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        ev0.record(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            go(i); evs[i].record(slots[i % depth][1])
        torch.cuda.synchronize()
        static = (time.perf_counter() - t0) * 1e3 / steps          # a fixed number of steps per slot: as slow as the most crowded hardware queue
        rate = bench.steady_rate_ms([ev0.elapsed_time(e) for e in evs], depth)   # what a rank that refills free slots sees
        return rate, int(sum(int(s[3].sum()) for s in slots) / depth), static
    for name, pick in (("contiguous", lambda r: d[r * m:(r + 1) * m]), ("interleaved", lambda r: d[r::N].contiguous())):
        if only and name != only: continue
        res = [run(pick(r)) for r in range(N)]
        ts = [x[0] for x in res]
        print("%d shards, %d steps in flight, %s: ms per step at the steady rate %s | max %.2f mean %.2f | lines %d | fixed steps per slot: %s" % (
            N, depth, name, " ".join("%.1f" % t for t in ts), max(ts), sum(ts) / N, sum(x[1] for x in res), " ".join("%.1f" % x[2] for x in res)), flush=True)
    for sl in slots: sl[0].close()
    del slots
    torch.cuda.empty_cache()
