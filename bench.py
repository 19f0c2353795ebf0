#!/usr/bin/env python3
"""bench.py -- the headline benchmark: whole-pipeline LSD throughput (Mpixels/s and lines/s) on a batch of
512 synthetic 2048x2048 occupancy maps per GPU (BASELINE.json configs[3]/[4], SURVEY 8d "C4").

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One process per GPU.  A "step" is one pass of the hot path (remap -> Gaussian -> gradient -> sort -> region
grow/rectangle/NFA -> line list + raster) over this rank's 512-image batch, inputs resident in HBM.  Images
are independent, so ranks never exchange data on the path (weak scaling: 512 images per GPU); for N > 1 each
step ends with the RCCL gather of the ragged line lists to rank 0.  Rank 0 prints ONE JSON line.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SOURCES = ["aisle1", "aisle2", "aisle3", "mapValue"]   # the four aisle-class fixtures
HBM_PEAK_GBS = 8000.0                                   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
GRAD_BYTES_PER_PX = 25.0                                # 8 B read + 8+8+1 B written per scaled pixel


def make_image(maps, i, size):
    """Image i of the synthetic batch: tile + roll + flip of one of four fixtures, seeded by i."""
    src = maps[SOURCES[i % 4]]
    hs, ws = src.shape
    canvas = np.tile(src, (-(-size // hs) + 1, -(-size // ws) + 1))
    if i == 0:
        dy = dx = 0
    else:
        rng = np.random.default_rng(1234 + i)
        dy, dx = (int(v) for v in rng.integers(0, [hs, ws]))
    canvas = np.roll(canvas, (dy, dx), (0, 1))[:size, :size]
    f = (i // 4) % 4
    if f & 1:
        canvas = canvas[:, ::-1]
    if f & 2:
        canvas = canvas[::-1, :]
    return np.ascontiguousarray(canvas)


def make_batch(maps, n, size, first=0):
    out = np.empty((n, size, size), np.uint8)
    for j in range(n):
        out[j] = make_image(maps, first + j, size)
    return out


def load_maps():
    z = np.load(os.path.join(ROOT, "tests", "golden", "maps.npz"))
    return {k: z[k] for k in z.files}


def cpu_baseline(maps, size, first, budget_s=15.0, max_images=512):
    """The CPU oracle (a single-threaded port of the reference path) on a bounded sample of the SAME workload."""
    from oracle import oracle
    oracle.build()
    oracle.lsd(maps["map1"].copy())   # warm-up
    t_used, px, nl, k = 0.0, 0, 0, 0
    while k < max_images and t_used < budget_s:
        img = make_image(maps, first + k, size)
        t0 = time.perf_counter()
        r = oracle.lsd(img, want_lineim=True)
        t_used += time.perf_counter() - t0
        px += size * size
        nl += len(r["lines"])
        k += 1
    return {"value": px / 1e6 / t_used, "unit": "Mpix/s", "cores": 1, "kind": "port",
            "sample": "first %d images of the same %dx%d batch, oracle/lsd_oracle.c single thread, %.1f s" % (k, size, size, t_used),
            "lines_per_s": nl / t_used}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=512, help="images per GPU")
    ap.add_argument("--size", type=int, default=2048)
    ap.add_argument("--max-lines", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip everything outside the timed region (CPU baseline, single-image latency, copy ceiling): what the profiling passes use")
    ap.add_argument("--no-lineim", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="exercise the RCCL gather path even with one rank (testing)")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (a.gpus, world, a.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or (a.force_dist and "RANK" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)    # RCCL

    lsd = importlib.import_module("linesegmentdetector-slam_amd")
    ldist = importlib.import_module("linesegmentdetector-slam_amd.dist")
    ctx = lsd.Context(local)
    maps = load_maps()

    n, size = a.batch, a.size
    n_total = n * world
    first = rank * n                                       # weak scaling: every GPU gets its own 512 images
    host = make_batch(maps, n, size, first)
    d_maps = torch.from_numpy(host).to(dev)
    del host
    d_lines = torch.zeros((n, a.max_lines, 10), dtype=torch.int64, device=dev)
    d_counts = torch.zeros(n, dtype=torch.int32, device=dev)
    d_ims = None if a.no_lineim else torch.zeros((n, size, size), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    ctx.reserve(n, size, size)
    w, h = lsd.scaled_size(size, size)
    kt = {k: 0.0 for k in ("gauss", "gradient", "sort", "region", "lines", "total")}

    def step(collect):
        ctx.enqueue_device(d_maps.data_ptr(), n, size, size, d_lines.data_ptr(), a.max_lines, d_counts.data_ptr(),
                           d_line_ims=None if d_ims is None else d_ims.data_ptr(), stream=stream)
        if use_dist:
            res = ldist.gather_line_lists(d_lines, d_counts, n_total, dst=0)
        else:
            res = None
        if collect:                                        # HIP events recorded on the launch stream by the library
            for k, v in ctx.timings().items():
                kt[k] += v
        return res

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    res = None
    for _ in range(a.steps):
        res = step(True)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    nl = d_counts.sum().to(torch.float64).reshape(1)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(nl, op=dist.ReduceOp.SUM)
    dt = float(tmax.item())
    total_lines = float(nl.item())
    overflow = int((d_counts > a.max_lines).sum().item())

    if rank == 0:
        if use_dist:
            offsets, lines = res
            assert int(offsets[-1]) == int(total_lines) and lines.shape[0] == int(total_lines)
        ms_per_step = dt / a.steps * 1e3
        mpix = n_total * size * size / 1e6
        value = mpix / (dt / a.steps)
        grad_ms = kt["gradient"] / a.steps
        grad_bytes = GRAD_BYTES_PER_PX * w * h * n
        achieved = grad_bytes / (grad_ms * 1e-3) / 1e9 if grad_ms > 0 else 0.0
        traffic, traffic_all = None, {}
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("k_gradient_bytes_per_launch")
                traffic_all = {k: v["hbm_bytes_per_launch"] for k, v in tj.get("kernels", {}).items()}
            except Exception:
                traffic = None
        out = {
            "metric": "Mpixels/sec LSD (grad+grow+NFA)", "value": value, "unit": "Mpix/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d x %dx%d u8 occupancy maps per GPU tiled/rolled/flipped from 4 aisle-class fixtures "
                                   "(SURVEY 8d C4), full pipeline incl. lineIm, params 0.3/0.6/22.5/0.7/1024" % (n, size, size),
                       "images_per_gpu": n, "image": [size, size], "scaled": [w, h],
                       "parallelism": "image-sharded x%d, RCCL gather of line lists" % world if world > 1 else "single GPU"},
            "lines_per_s": total_lines / (dt / a.steps), "lines_per_step": total_lines, "line_overflow_images": overflow,
            "kernel_ms": {k: v / a.steps for k, v in kt.items()},
            # informational: every kernel's HBM traffic (PMC, profiles/traffic_latest.json) over its live launch time
            "kernel_hbm_GBs": {k: traffic_all["k_" + k] / (kt[k] / a.steps * 1e-3) / 1e9
                               for k in ("gauss", "gradient", "sort", "region", "lines")
                               if n == 512 and size == 2048 and ("k_" + k) in traffic_all and kt[k] > 0},
            "roofline": {"kernel": "k_gradient", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": grad_bytes, "avg_launch_ms": grad_ms,
                         "note": "north_star prices the gradient pass; k_region takes 97% of the step but is a serial latency chain "
                                 "(about 2% of HBM peak, no MFMA work): DESIGN.md section 4, kernel_hbm_GBs below"},
        }
        if world == 1 and not a.no_cpu_baseline:
            # extras outside the timed region: single-image latency, and the device-copy ceiling of this box
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            lat = []
            for _ in range(3):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                ctx.enqueue_device(d_maps.data_ptr(), 1, size, size, d_lines.data_ptr(), a.max_lines, d_counts.data_ptr(),
                                   d_line_ims=None if d_ims is None else d_ims.data_ptr(), stream=stream)
                torch.cuda.synchronize()
                lat.append((time.perf_counter() - t1) * 1e3)
            out["single_image_latency_ms"] = min(lat)
            src_t = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
            dst_t = torch.empty_like(src_t)
            dst_t.copy_(src_t)
            ev0.record()
            for _ in range(5):
                dst_t.copy_(src_t)
            ev1.record()
            torch.cuda.synchronize()
            out["roofline"]["measured_copy_GBs"] = 5 * 2 * (1 << 30) / (ev0.elapsed_time(ev1) * 1e-3) / 1e9   # read + write
            del src_t, dst_t
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(maps, size, first)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
