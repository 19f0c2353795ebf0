#!/usr/bin/env python3
"""Developer probe: the timed configuration (D steps in flight, 4-wave region stage, help off) with every slot's stream confined to a share
of the CUs (hipExtStreamCreateWithCUMask): a static partition of the GPU instead of the dispatcher's.
   tools/cumask_probe.py <layout> [depth [steps]]      layout: none | block | stride | block2 (two slots share a 64-CU block)"""
import ctypes as C, importlib, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
lsd = importlib.import_module("linesegmentdetector-slam_amd")
layout = sys.argv[1] if len(sys.argv) > 1 else "none"
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 8
K = int(sys.argv[3]) if len(sys.argv) > 3 else 64
hip = C.CDLL("libamdhip64.so")
torch.cuda.init(); torch.zeros(1, device="cuda")
ncu = torch.cuda.get_device_properties(0).multi_processor_count
def make_stream(j):
    if layout == "none":
        return torch.cuda.Stream().cuda_stream
    bits = np.zeros(ncu, bool)
    if layout == "block": bits[j * ncu // depth:(j + 1) * ncu // depth] = True
    elif layout == "stride": bits[j::depth] = True
    elif layout == "block2": g = depth // 2; bits[(j % g) * ncu // g:((j % g) + 1) * ncu // g] = True
    else: raise SystemExit("layout?")
    words = np.zeros((ncu + 31) // 32, np.uint32)
    for i in np.nonzero(bits)[0]: words[i >> 5] |= np.uint32(1 << (i & 31))
    s = C.c_void_p()
    r = hip.hipExtStreamCreateWithCUMask(C.byref(s), C.c_uint32(len(words)), words.ctypes.data_as(C.c_void_p))
    assert r == 0, r
    return s.value
maps = bench.load_maps(); n, size = 512, 2048
sub = int(os.environ.get("SUB", "1"))          # launches per step: the batch in `sub` parts, each on its own slot
n //= sub
d_all = torch.from_numpy(bench.make_batch(maps, n * sub, size)).cuda()
ctxs = [lsd.Context(0) for _ in range(depth)]
for c in ctxs:
    c.set_region_waves(4); c.set_region_help(0); c.reserve(n, size, size)
streams = [make_stream(j) for j in range(depth)]
outs = [(torch.zeros((n, 1024, 10), dtype=torch.int64, device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda"),
         torch.zeros((n, size, size), dtype=torch.uint8, device="cuda")) for _ in range(depth)]
def step(i, nn=n):
    j = i % depth
    l, c, im = outs[j]
    ctxs[j].enqueue_device(d_all[(i % sub) * n:].data_ptr(), nn, size, size, l.data_ptr(), 1024, c.data_ptr(), d_line_ims=im.data_ptr(), stream=streams[j])
for j in range(depth): step(j, 1)
torch.cuda.synchronize()
for i in range(2 * depth): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(K): step(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K * sub
ok = sum(int(o[1].sum()) for o in outs[:sub]) == 138815 if depth % sub == 0 else None
tm = ctxs[(K - 1) % depth].timings()
print("parts %d" % sub, "layout %s depth %d: %.2f ms per step over %d steps; lines ok %s; last step's events: %s" % (layout, depth, dt * 1e3, K, ok, " ".join("%s %.1f" % kv for kv in tm.items())), flush=True)
