#!/usr/bin/env python3
"""Developer probe: repeats a batch and reports images the region stage's watchdog gave up on (counts == -1) with the state it
recorded; also checks that the line records are identical from run to run.   tools/hang_probe.py [n] [size] [reps] [waves]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
lsd = importlib.import_module("linesegmentdetector-slam_amd")
a = [int(x) for x in sys.argv[1:]]
n, size, reps, waves = (a + [48, 1024, 5, 0][len(a):])[:4]
maps = bench.load_maps(); ctx = lsd.Context(0); ctx.set_region_waves(waves)
d = torch.from_numpy(bench.make_batch(maps, n, size)).cuda()
lines = torch.zeros((n, 1024, 10), dtype=torch.int64, device="cuda"); counts = torch.zeros(n, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
wh = lsd.scaled_size(size, size)
ref = None
for rep in range(reps):
    lines.zero_()
    t0 = time.perf_counter()
    ctx.enqueue_device(d.data_ptr(), n, size, size, lines.data_ptr(), 1024, counts.data_ptr(), stream=s); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c = counts.cpu().numpy()
    bad = np.nonzero(c < 0)[0]
    print("rep", rep, "%.1f ms" % (dt * 1e3), "region %.1f" % ctx.timings()["region"], "lines", int(c[c > 0].sum()), "aborted", bad.tolist(), flush=True)
    for i in bad[:4]:
        v = np.zeros(48, np.int64)                           # the raw stats record: the watchdog's words are 40..47, per-wave words 24..39
        assert ctx.L.lsd_debug_fetch(ctx.h, int(i), lsd.DBG_STATS, v.ctypes.data, v.nbytes) == 0
        v = [int(x) for x in v]
        print("   image", i, dict(zip(("s_commit", "s_next", "nseeds", "state_at_cursor", "s_nbig", "s_lock", "pend_k", "wave"), v[40:48])), flush=True)
        print("      per wave (chunk start, pend_k, ch_pend):", [(x & 0xffffffff, (x >> 32) - 1, hex(y)) for x, y in zip(v[24:40:2], v[25:40:2])], flush=True)
    if os.environ.get("HELPSTATS"):
        st = [ctx.fetch(i, lsd.DBG_STATS, wh) for i in range(n)]
        tot = np.array([x["cycles_total"] for x in st]) / 1e6
        t0s = np.array([x["wd_pend"] for x in st], np.float64); t1s = np.array([x["wd_wave"] for x in st], np.float64)
        # (the counter is per XCD; every XCD has workgroups that started with the launch)
        cl = np.array([x["wd_lock"] for x in st])          # XCC id
        b0, e0 = (t0s - t0s.min()) / 1e5, (t1s - t0s.min()) / 1e5      # ms (s_memrealtime: 100 MHz)
        print("   timeline (ms from the first start): last start %.1f, ends: median %.1f p90 %.1f max %.1f" % (b0.max(), np.median(e0), np.percentile(e0, 90), e0.max()))
        for i in np.argsort(-e0)[:8]: print("      image %d: start %.1f end %.1f Mcycles %.0f exports %d" % (i, b0[i], e0[i], tot[i], st[i]["help_exports"]))
        print("      started after 0.1 ms: %d images; their starts: p10 %.1f median %.1f p90 %.1f" % ((b0 > 0.1).sum(), *np.percentile(b0[b0 > 0.1], [10, 50, 90])))
        print("      per XCD: images, last start, last end:", [(int((cl == x).sum()), int(b0[cl == x].max()), int(e0[cl == x].max())) for x in sorted(set(cl.tolist()))])
        print("   help: exports %d evals %d | Mcycles per image: mean %.0f max %.0f" % (sum(x["help_exports"] for x in st), sum(x["help_evals"] for x in st), tot.mean(), tot.max()), flush=True)
    cur = (c.tobytes(), lines.cpu().numpy().tobytes())
    if ref is None: ref = cur
    elif cur != ref: print("   DIFFERS from run 0", flush=True)
