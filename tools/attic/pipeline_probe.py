#!/usr/bin/env python3
"""Developer probe: throughput of the bench batch when consecutive steps overlap (D contexts on D streams, step i on slot i % D):
the tail of one step's region stage -- a few images on a CU each -- runs next to the head of the next one's.
   tools/pipeline_probe.py [depth ...]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
lsd = importlib.import_module("linesegmentdetector-slam_amd")
maps = bench.load_maps(); n, size = 512, 2048
d = torch.from_numpy(bench.make_batch(maps, n, size)).cuda()
for depth in [int(a) for a in sys.argv[1:]] or [1, 2, 3]:
    ctxs = [lsd.Context(0) for _ in range(depth)]
    for c_ in ctxs: c_.set_region_waves(int(os.environ.get("WAVES", "0")))
    streams = [torch.cuda.Stream() for _ in range(depth)]
    outs = [(torch.zeros((n, 1024, 10), dtype=torch.int64, device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda"), torch.zeros((n, size, size), dtype=torch.uint8, device="cuda")) for _ in range(depth)]
    def step(i):
        j = i % depth
        l, c, im = outs[j]
        ctxs[j].enqueue_device(d.data_ptr(), n, size, size, l.data_ptr(), 1024, c.data_ptr(), d_line_ims=im.data_ptr(), stream=streams[j].cuda_stream)
    for i in range(depth): step(i)
    torch.cuda.synchronize()
    K = int(os.environ.get("K", "12"))
    t0 = time.perf_counter()
    enq = []
    for i in range(K):
        te = time.perf_counter(); step(i); enq.append((time.perf_counter() - te) * 1e3)
    t_enq = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    if os.environ.get("ENQ"): print("   host: all %d enqueues took %.1f ms; per call: %s" % (K, t_enq, " ".join("%.0f" % x for x in enq)))
    dt = (time.perf_counter() - t0) / K
    ok = all(int(o[1].sum()) == 138815 for o in outs)
    tm = ctxs[0].timings()
    wh = lsd.scaled_size(size, size)
    if os.environ.get("TIMELINE"):
        # the last launch of every slot (the final `depth` steps: the first of them started in the steady state)
        rows = []
        for j in range(depth):
            st = [ctxs[j].fetch(i, lsd.DBG_STATS, wh) for i in range(n)]
            t0s = np.array([x["wd_pend"] for x in st], float) / 1e5; t1s = np.array([x["wd_wave"] for x in st], float) / 1e5
            cyc = np.array([x["cycles_total"] for x in st]) / 1e6
            rows.append((t0s.min(), j, t0s, t1s, cyc))
        base = min(r[0] for r in rows)
        for t0min, j, t0s, t1s, cyc in sorted(rows):
            print("   slot %d: first start %.1f ms, starts p50 %.1f p90 %.1f last %.1f | ends p50 %.1f p90 %.1f last %.1f | Mcycles mean %.0f max %.0f" % (
                j, t0min - base, *(np.percentile(t0s, [50, 90]) - base), t0s.max() - base, *(np.percentile(t1s, [50, 90]) - base), t1s.max() - base, cyc.mean(), cyc.max()))
    print("depth %d: %.1f ms per step = %.1f Gpix/s; lines ok %s; last launch on slot 0: gauss %.2f gradient %.2f sort %.2f region %.1f" % (depth, dt * 1e3, n * size * size / dt / 1e9, ok, tm["gauss"], tm["gradient"], tm["sort"], tm["region"]), flush=True)
    del ctxs, outs
    torch.cuda.empty_cache()
