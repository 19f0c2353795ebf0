#!/usr/bin/env python3
"""Developer probe (run on the GPU box, CPU only): why does the all-cores figure of bench.py's cpu_baseline collapse per core?
Runs k pinned single-thread oracle instances (k = 1, 8, 32, 64, 128, all) over 4 bench images each, with glibc's default
allocator behaviour (every call mmaps and page-faults ~30 MB of work arrays) and with the heap kept (mallopt), and prints the
slowest / median instance time next to the host's topology and cgroup CPU quota."""
import multiprocessing as mp, os, statistics, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


def worker(args):
    core, first, count, keep_heap = args
    os.sched_setaffinity(0, {core})
    if keep_heap:
        bench.keep_heap()
    from oracle import oracle
    maps = bench.load_maps()
    oracle.lsd(maps["map1"].copy())
    imgs = [bench.make_image(maps, first + k, 2048) for k in range(count)]
    ts = []
    for rep in range(2):
        t0 = time.perf_counter()
        for im in imgs:
            oracle.lsd(im.copy(), want_lineim=True)
        ts.append(time.perf_counter() - t0)
    return ts[1]


if __name__ == "__main__":
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
        if os.path.exists(f):
            print(f, open(f).read().strip())
    print(subprocess.run("lscpu | egrep 'Model name|Socket|Core|Thread|NUMA|MHz|^CPU\\(s\\)'; nproc; free -g | head -2", shell=True, capture_output=True, text=True).stdout)
    cores = sorted(os.sched_getaffinity(0))
    print("affinity:", len(cores), "cpus")
    # physical cores: one sibling per core
    sib = {}
    for c in cores:
        try:
            s = open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read().strip()
        except OSError:
            s = str(c)
        sib.setdefault(s, c)
    phys = sorted(sib.values())
    print("physical cores (one thread each):", len(phys))
    from oracle import oracle
    oracle.build()
    for keep in (0, 1):
        for k in (1, 8, 32, 64, 128, len(phys), len(cores)):
            use = (phys if k <= len(phys) else cores)[:k]
            with mp.get_context("spawn").Pool(len(use)) as pool:
                ts = pool.map(worker, [(c, 100 + 4 * j, 4, keep) for j, c in enumerate(use)])
            px = 4 * 2048 * 2048 / 1e6
            print("keep_heap %d instances %3d: slowest %.2f s median %.2f s  -> %.1f Mpix/s per instance (slowest), %.0f Mpix/s aggregate" % (
                keep, len(use), max(ts), statistics.median(ts), px / max(ts), len(use) * px / max(ts)), flush=True)
