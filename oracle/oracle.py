"""ctypes front-end of the CPU ORACLE (oracle/lsd_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package never does.  The shared object is built by `make -C oracle`
(__graft_entry__.build() does that).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OrcLine(C.Structure):  # == structLinesInfo, LSD/baseFunc.h:33-44
    _fields_ = [(n, C.c_double) for n in ("k", "b", "dx", "dy", "x1", "y1", "x2", "y2", "len")] + [("orient", C.c_int)]


LINE_DTYPE = np.dtype([("k", "f8"), ("b", "f8"), ("dx", "f8"), ("dy", "f8"), ("x1", "f8"), ("y1", "f8"),
                       ("x2", "f8"), ("y2", "f8"), ("len", "f8"), ("orient", "i4"), ("_pad", "i4")])
assert LINE_DTYPE.itemsize == 80 == C.sizeof(OrcLine)


class OrcSeed(C.Structure):
    _fields_ = [("order_idx", C.c_int), ("x", C.c_int), ("y", C.c_int), ("num", C.c_int),
                ("outcome", C.c_int), ("final_num", C.c_int), ("logNFA", C.c_double)]


SEED_DTYPE = np.dtype([("order_idx", "i4"), ("x", "i4"), ("y", "i4"), ("num", "i4"),
                       ("outcome", "i4"), ("final_num", "i4"), ("logNFA", "f8")])
assert SEED_DTYPE.itemsize == C.sizeof(OrcSeed)


class OrcDebug(C.Structure):
    _fields_ = [("w", C.c_int), ("h", C.c_int),
                ("gauss", C.POINTER(C.c_double)), ("mag", C.POINTER(C.c_double)), ("deg", C.POINTER(C.c_double)),
                ("used0", C.POINTER(C.c_uint8)), ("used", C.POINTER(C.c_uint8)),
                ("nb", C.c_int),
                ("ord_v", C.POINTER(C.c_int)), ("ord_x", C.POINTER(C.c_int)), ("ord_y", C.POINTER(C.c_int)),
                ("maxGrad", C.c_double),
                ("n_seed", C.c_int), ("seeds", C.POINTER(OrcSeed)),
                ("recs", C.POINTER(C.c_double)),
                ("grow_calls", C.c_long), ("grown_px", C.c_long), ("nfa_calls", C.c_long),
                ("rrr_calls", C.c_long), ("rrr_passes", C.c_long), ("rrr_sentinel_drops", C.c_long),
                ("rrr_oob_reads", C.c_long), ("nfa_min_abs", C.c_double), ("nfa_min_gap", C.c_double)]


def build(asan=False):
    target = "liblsd_oracle_asan.so" if asan else "liblsd_oracle.so"
    subprocess.run(["make", "-C", _HERE, target], check=True, capture_output=True)
    return os.path.join(_HERE, target)


def lib_cr():
    """The diagnostic variant with correctly rounded sin/cos/atan2 (oracle/cr_shim.cpp); built on demand."""
    p = os.path.join(_HERE, "liblsd_oracle_cr.so")
    if not os.path.exists(p) or os.path.getmtime(p) < os.path.getmtime(os.path.join(_HERE, "lsd_oracle.c")):
        subprocess.run(["make", "-C", _HERE, "liblsd_oracle_cr.so"], check=True, capture_output=True)
    return lib(p)


def lib(path=None):
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = path or os.path.join(_HERE, "liblsd_oracle.so")
    if not os.path.exists(p):
        build()
    L = C.CDLL(p)
    L.orc_lsd.restype = C.c_int
    L.orc_lsd.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_double, C.c_double, C.c_double,
                          C.c_double, C.c_int, C.c_void_p, C.POINTER(C.POINTER(OrcLine)), C.POINTER(C.c_int),
                          C.POINTER(OrcDebug)]
    L.orc_map_cache.restype = C.c_int
    L.orc_map_cache.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_double, C.c_double, C.c_void_p]
    L.orc_selftest_qsort_stable.restype = C.c_int
    L.orc_selftest_qsort_stable.argtypes = [C.c_int, C.c_uint]
    L.orc_log_gamma.restype = C.c_double
    L.orc_log_gamma.argtypes = [C.c_int]
    L.orc_free.argtypes = [C.c_void_p]
    L.orc_debug_free.argtypes = [C.POINTER(OrcDebug)]
    if path is None:
        _LIB = L
    return L


DEFAULTS = dict(sca=0.3, sig=0.6, angThre=22.5, denThre=0.7, pseBin=1024)  # LSD/baseFunc.h:64-68


def _arr(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


def lsd(map_u8, sca=0.3, sig=0.6, angThre=22.5, denThre=0.7, pseBin=1024, want_lineim=True, debug=False,
        _lib=None):
    """Runs the oracle on a COPY-FREE uint8 image (it is mutated in place, like the reference does).

    Returns dict(lines=structured array[LINE_DTYPE], lineIm=uint8[rows,cols] or None, dbg=dict or None).
    """
    L = _lib or lib()
    assert map_u8.dtype == np.uint8 and map_u8.ndim == 2 and map_u8.flags.c_contiguous
    rows, cols = map_u8.shape
    line_im = np.zeros((rows, cols), np.uint8) if want_lineim else None
    lines_p = C.POINTER(OrcLine)()
    n = C.c_int(0)
    dbg = OrcDebug() if debug else None
    rc = L.orc_lsd(map_u8.ctypes.data, cols, rows, map_u8.strides[0], sca, sig, angThre, denThre, pseBin,
                   line_im.ctypes.data if want_lineim else None, C.byref(lines_p), C.byref(n),
                   C.byref(dbg) if debug else None)
    if rc != 0:
        raise RuntimeError("orc_lsd failed: %d" % rc)
    lines = np.zeros(n.value, LINE_DTYPE)
    if n.value:
        C.memmove(lines.ctypes.data, lines_p, 80 * n.value)
        lines["_pad"] = 0
    L.orc_free(lines_p)
    out = {"lines": lines, "lineIm": line_im, "dbg": None}
    if debug:
        w, h = dbg.w, dbg.h
        npx = w * h
        d = {"w": w, "h": h, "nb": dbg.nb, "maxGrad": dbg.maxGrad}
        if npx:
            d["gauss"] = _arr(dbg.gauss, npx, np.float64).reshape(h, w)
            d["mag"] = _arr(dbg.mag, npx, np.float64).reshape(h, w)
            d["deg"] = _arr(dbg.deg, npx, np.float64).reshape(h, w)
            d["used0"] = _arr(dbg.used0, npx, np.uint8).reshape(h, w)
            d["used"] = _arr(dbg.used, npx, np.uint8).reshape(h, w)
            d["ord_v"] = _arr(dbg.ord_v, dbg.nb, np.int32)
            d["ord_x"] = _arr(dbg.ord_x, dbg.nb, np.int32)
            d["ord_y"] = _arr(dbg.ord_y, dbg.nb, np.int32)
            seeds = np.zeros(dbg.n_seed, SEED_DTYPE)
            if dbg.n_seed:
                C.memmove(seeds.ctypes.data, dbg.seeds, SEED_DTYPE.itemsize * dbg.n_seed)
            d["seeds"] = seeds
            d["recs"] = _arr(dbg.recs, 12 * n.value, np.float64).reshape(n.value, 12)
        for k in ("grow_calls", "grown_px", "nfa_calls", "rrr_calls", "rrr_passes", "rrr_sentinel_drops",
                  "rrr_oob_reads", "nfa_min_abs", "nfa_min_gap"):
            d[k] = getattr(dbg, k)
        L.orc_debug_free(C.byref(dbg))
        out["dbg"] = d
    return out


def map_cache(map_u8, res, z_occ_max_dis=1.0, _lib=None):
    """Oracle for mylsd::createMapCache (LSD/myLSD.cpp:11-127); z_occ_max_dis from baseFunc.h:60."""
    L = _lib or lib()
    assert map_u8.dtype == np.uint8 and map_u8.ndim == 2 and map_u8.flags.c_contiguous
    rows, cols = map_u8.shape
    out = np.zeros((rows, cols), np.float64)
    rc = L.orc_map_cache(map_u8.ctypes.data, cols, rows, map_u8.strides[0], res, z_occ_max_dis, out.ctypes.data)
    if rc != 0:
        raise RuntimeError("orc_map_cache failed: %d" % rc)
    return out


def occupancy_to_map(grid_i8):
    """Oracle for the cell loop of mapCallback (LSD/main_on_linux.cpp:108-124): the int8 cell read as uint8 is mapped
    255 -> 0, 0 -> 255, anything else -> 1.  PARITY UNPINNED: the reference ships no input/output pair for this loop
    (data/mapValue.txt is an output only); the restatement is three branches read off the source."""
    v = np.ascontiguousarray(grid_i8).view(np.uint8)
    out = np.ones(v.shape, np.uint8)
    out[v == 255] = 0
    out[v == 0] = 255
    return out


class _Pos(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("ang", C.c_double)]


def scan_to_map_match(map_cache, map_lines, scan_lines, scan_im_points, lidar_pose, last_pose, pairs, z_occ=1.0,
                      max_esti_dist=60.0, _lib=None):
    """Oracle for one batch of myfa::thread_ScanToMapMatch (LSD/myFA.cpp:197-396).  PARITY UNPINNED (lsd_oracle.h).
    Returns float64 [m, 4, 4]: (x, y, ang, score) per pair and matching."""
    L = _lib or lib()
    L.orc_scan_to_map_match.restype = C.c_int
    L.orc_scan_to_map_match.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                        _Pos, _Pos, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p]
    mc = np.ascontiguousarray(map_cache, np.float64)
    rows, cols = mc.shape
    ml = np.ascontiguousarray(map_lines); sl = np.ascontiguousarray(scan_lines)
    assert ml.dtype.itemsize == 80 and sl.dtype.itemsize == 80
    pts = np.ascontiguousarray(scan_im_points).view(np.float64).reshape(-1, 3)
    pr = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
    out = np.zeros((len(pr), 4, 4), np.float64)
    mk = lambda p: _Pos(float(p[0]), float(p[1]), float(p[2]) if len(p) > 2 else 0.0)
    rc = L.orc_scan_to_map_match(mc.ctypes.data, cols, rows, ml.ctypes.data, len(ml), sl.ctypes.data, len(sl), pts.ctypes.data,
                                 len(pts), mk(lidar_pose), mk(last_pose), pr.ctypes.data, len(pr), z_occ, max_esti_dist, out.ctypes.data)
    if rc != 0:
        raise RuntimeError("orc_scan_to_map_match failed: %d" % rc)
    return out


class _Polar(C.Structure):
    _fields_ = [("range", C.c_double), ("angle", C.c_double)]


class _MapParam(C.Structure):
    _fields_ = [("oriMapCol", C.c_int), ("oriMapRow", C.c_int), ("mapResol", C.c_double), ("mapOriX", C.c_double), ("mapOriY", C.c_double)]


def feature_scan(scan, map_param, region_point_limit=3, thre_line=0.08, line_dist_thre_m=0.5, pts_cap=8192, _lib=None):
    """Oracle for myrdp::FeatureScan (LSD/myRDP.cpp:9-185) on ONE scan: scan float64 [len_lp, 2] = (range, angle) of the finite
    readings, map_param = (oriMapCol, oriMapRow, mapResol, mapOriX, mapOriY); defaults = baseFunc.h:70-72.
    Returns dict(lines LINE_DTYPE[n], pts float64 [m, 3], lidar_pos (x, y), im_size (oriXLim, oriYLim), lineIm uint8 [oriYLim, oriXLim])."""
    L = _lib or lib()
    L.orc_feature_scan.restype = C.c_int
    L.orc_feature_scan.argtypes = [_MapParam, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p, C.POINTER(C.c_int),
                                   C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_void_p]
    sc = np.ascontiguousarray(scan, np.float64).reshape(-1, 2)
    lines = np.zeros(360, LINE_DTYPE)
    pts = np.zeros((pts_cap, 3), np.float64)
    nl, npts = C.c_int(0), C.c_int(0)
    lidar = np.zeros(2, np.float64); size = np.zeros(2, np.int32)
    mp = _MapParam(int(map_param[0]), int(map_param[1]), float(map_param[2]), float(map_param[3]), float(map_param[4]))
    rc = L.orc_feature_scan(mp, sc.ctypes.data, len(sc), region_point_limit, thre_line, line_dist_thre_m, lines.ctypes.data, C.byref(nl),
                            pts.ctypes.data, pts_cap, C.byref(npts), lidar.ctypes.data, size.ctypes.data)
    if rc != 0:
        raise RuntimeError("orc_feature_scan failed: %d" % rc)
    assert npts.value <= pts_cap                      # (nl may pass 360: the first 360 records are stored, as the HIP entry point does)
    pts = pts[:npts.value].copy()
    im = np.zeros((max(int(size[1]), 0), max(int(size[0]), 0)), np.uint8)
    if len(pts):
        im[pts[:, 1].astype(int), pts[:, 0].astype(int)] = 255
    return dict(lines=lines[:min(nl.value, 360)].copy(), n_lines=nl.value, pts=pts, lidar_pos=(float(lidar[0]), float(lidar[1])), im_size=(int(size[0]), int(size[1])), lineIm=im)
