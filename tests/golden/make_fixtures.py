#!/usr/bin/env python3
"""Pack the reference's map fixtures into tests/golden/maps.npz (run in the build container only).

The fixture is DATA: the occupancy-grid text files the reference's replay driver loads
(/root/reference/data*/mapValue*.txt + mapParam*.txt) converted with the driver's own
loading convention (LSD/main_on_windows.cpp:27-46):
  * mapParam order is cols, rows, resolution, originX, originY (main_on_windows.cpp:32);
  * every token is read with "%d" into a uint8 cell, i.e. the low byte is kept (-1 -> 255).
Also packs the reference's MATLAB-era golden outputs for data/mapValue.txt
(data/MaplinesInfo.txt: 40x10 line table; data/MaplineIm.txt: lit pixels of the 428x1377
raster) which serve as a LOOSE sanity oracle.

/root/reference does not exist on the GPU box; only the .npz travels.
"""
import os, sys, json
import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "maps.npz")

MAPS = {
    # name: (value file, param file)
    "map1":   ("data/mapValue_map1.txt",   "data/mapParam_map1.txt"),
    "map2":   ("data/mapValue_map2.txt",   "data/mapParam_map1.txt"),
    "mapValue": ("data/mapValue.txt",      "data/mapParam.txt"),
    "aisle1": ("data/mapValue_aisle1.txt", "data/mapParam_aisle1.txt"),
    "aisle2": ("data/mapValue_aisle2.txt", "data/mapParam_aisle2.txt"),
    "aisle3": ("data/mapValue_aisle3.txt", "data/mapParam_aisle3.txt"),
    "f3key":  ("data_20190513/data_f3key/data1/mapValue.txt", "data_20190513/data_f3key/data1/mapParam.txt"),
    "f4key":  ("data_20190514/data_f4key/data1/mapValue.txt", "data_20190514/data_f4key/data1/mapParam.txt"),
}

def load_map(vfile, pfile):
    p = open(os.path.join(REF, pfile)).read().split()
    cols, rows, res = int(p[0]), int(p[1]), float(p[2])
    toks = np.array(open(os.path.join(REF, vfile)).read().split(), dtype=np.int64)
    assert toks.size == rows * cols, (vfile, toks.size, rows, cols)
    img = (toks & 0xFF).astype(np.uint8).reshape(rows, cols)   # "%d" into uint8 keeps the low byte
    return img, res

def main():
    out = {}
    meta = {}
    for name, (vf, pf) in MAPS.items():
        img, res = load_map(vf, pf)
        if name == "map2":
            assert np.array_equal(img, out["map1"]), "map2 is expected to be byte-identical to map1"
            continue  # not stored twice
        out[name] = img
        meta[name] = {"cols": int(img.shape[1]), "rows": int(img.shape[0]), "res": res,
                      "values": sorted(int(v) for v in np.unique(img))}
        print(name, img.shape, meta[name]["values"])
    li = np.loadtxt(os.path.join(REF, "data/MaplinesInfo.txt"))
    assert li.shape == (40, 10)
    out["matlab_MaplinesInfo"] = li
    im = np.loadtxt(os.path.join(REF, "data/MaplineIm.txt"))
    assert im.shape == (428, 1377)
    ys, xs = np.nonzero(im)
    out["matlab_MaplineIm_lit_yx"] = np.stack([ys, xs], 1).astype(np.int32)
    np.savez_compressed(OUT, **out)
    # the lidar log and the MATLAB prototype's scan lines (data/Lidar.txt: 99 frames x 360 (range, angle) readings, inf = no return;
    # data/ScanlinesInfo.txt / ScanlineIm.txt: its 12 lines and raster for frame 31 of that log, found by trying every frame)
    lid = np.loadtxt(os.path.join(REF, "data/Lidar.txt")).reshape(-1, 360, 2)
    sl = np.loadtxt(os.path.join(REF, "data/ScanlinesInfo.txt"))
    sim = np.loadtxt(os.path.join(REF, "data/ScanlineIm.txt"))
    assert lid.shape == (99, 360, 2) and sl.shape == (12, 9) and sim.shape == (237, 832)
    ys, xs = np.nonzero(sim)
    np.savez_compressed(os.path.join(os.path.dirname(OUT), "lidar.npz"), lidar=lid.astype(np.float64), matlab_ScanlinesInfo=sl,
                        matlab_ScanlineIm_lit_yx=np.stack([ys, xs], 1).astype(np.int32), matlab_ScanlineIm_shape=np.array(sim.shape, np.int32),
                        matlab_frame=np.int32(31))
    json.dump(meta, open(os.path.join(os.path.dirname(OUT), "maps_meta.json"), "w"), indent=1)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")

if __name__ == "__main__":
    main()
