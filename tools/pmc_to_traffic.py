#!/usr/bin/env python3
"""Derives profiles/traffic_latest.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs).

usage: pmc_to_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> [out.json [build tag]]
Counters are KB per dispatch.  HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: the factor 2 on FETCH_SIZE is
the gfx950 correction of MI355X_MICROARCH.md, confirmed for this code's 8-B-per-lane loads with tools/pmc_calib.py."""
import csv, glob, json, os, sys
from collections import defaultdict

# (the two builds of the region stage are kept apart: a run with several steps in flight uses w4 in its timed region and w8 in the
#  un-overlapped steps after it; "k_region" = whichever of the two has more launches in the pass, the configuration's own)
KERNELS = ("k_gauss", "k_gradient", "k_sort", "w4::k_region", "w8::k_region", "k_lines", "k_clear")


def per_kernel(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            for k in KERNELS:
                if k in row["Kernel_Name"]:
                    acc[k].append((int(row["Grid_Size"]), float(row["Counter_Value"])))
    # launches over the whole batch only: bench.py's set-up runs one IMAGE per context (first launch, first touch of the workspace),
    # and those launches must not dilute the average
    full = {k: [c for g, c in v if g == max(g2 for g2, _ in v)] for k, v in acc.items()}
    return {k: sum(v) / len(v) for k, v in full.items()}, {k: len(v) for k, v in full.items()}


def source_sha():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    return bench.source_sha()


def main():
    fd, wd = sys.argv[1], sys.argv[2]
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic_latest.json")
    build = sys.argv[4] if len(sys.argv) > 4 else "build not recorded"
    f, nf = per_kernel(fd, "FETCH_SIZE")
    w, nw = per_kernel(wd, "WRITE_SIZE")
    res = {"_method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --steps 1 --warmup 1 "
           "--no-cpu-baseline` (512 x 2048^2); counters are KB per dispatch, averaged over the whole-batch launches of each kernel (the one-image launches of the set-up are left out); "
           "FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (confirmed for 8-B-per-lane loads with "
           "tools/pmc_calib.py: profiles/r01b_pmc_calib_*.csv)",
           "build": build, "source_sha": source_sha(), "kernels": {}}
    for k in KERNELS:
        if k in f and k in w:
            res["kernels"][k] = {"FETCH_SIZE_KB": f[k], "WRITE_SIZE_KB": w[k], "launches": [nf[k], nw[k]],
                                 "hbm_bytes_per_launch": (2 * f[k] + w[k]) * 1024}
    reg = [k for k in ("w4::k_region", "w8::k_region") if k in res["kernels"]]
    if reg:
        main = max(reg, key=lambda k: res["kernels"][k]["launches"][0])
        res["kernels"]["k_region"] = dict(res["kernels"][main], variant=main[:2])
    res["k_gradient_bytes_per_launch"] = res["kernels"].get("k_gradient", {}).get("hbm_bytes_per_launch")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
