import importlib, os, sys, time, numpy as np
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/oracle") else os.getcwd())
lsd = importlib.import_module("linesegmentdetector-slam_amd")
ctx = lsd.Context(0)
yy, xx = np.mgrid[0:2000, 0:1500]
img = ((xx % 120) * 255 // 119).astype(np.uint8)
for waves in (8, 4):
    ctx.set_region_waves(waves)
    for rep in range(2):
        t = time.time(); lines, im = ctx.run(img.copy()); tg = time.time() - t
    st = ctx.fetch(0, lsd.DBG_STATS, lsd.scaled_size(1500, 2000))
    print("waves", waves, "run %.3f s region %.1f ms lines %d" % (tg, ctx.timings()["region"], len(lines)), {k: (round(v / 1e6, 1) if k.startswith("cycles") or k.startswith("wait") else v) for k, v in st.items() if v and not k.startswith("wd_") and not k.startswith("nfa_min")})
