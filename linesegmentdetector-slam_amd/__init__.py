"""linesegmentdetector-slam_amd -- MI355X-native LSD line-feature extractor.

Python host-side mirror of the reference interface for the hot path
(``mylsd::myLineSegmentDetector``, LSD/myLSD.h:132) on top of the C ABI of ``liblsdhip.so``
(include/lsd_hip.h).  The directory name contains a hyphen, import it with::

    lsd = importlib.import_module("linesegmentdetector-slam_amd")

There is NO CPU fallback: without the built HIP library, or without a GPU, every entry point
raises.  (The CPU oracle under oracle/ is test infrastructure and is never imported from here.)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblsdhip.so")

# ---- constants mirrored from include/lsd_hip.h -------------------------------------------------
LSD_OK, LSD_ERR_INVALID, LSD_ERR_NO_DEVICE, LSD_ERR_HIP, LSD_ERR_UNSUPPORTED, LSD_ERR_CAPACITY, LSD_ERR_NOMEM, LSD_ERR_INTERNAL = range(8)
LSD_FLAG_WRITEBACK_MAP = 1
STAGE_ALL, STAGE_GAUSS, STAGE_GRAD, STAGE_SORT, STAGE_REGION = range(5)
(DBG_GAUSS, DBG_MAG, DBG_DEG, DBG_STATE, DBG_ORDER, DBG_ORDER_VAL, DBG_NB, DBG_MAXGRAD, DBG_RECS, DBG_SEEDS,
 DBG_NSEED, DBG_STATS) = range(1, 13)

# LSD defaults, LSD/baseFunc.h:64-68
lsd_sca, lsd_sig, lsd_angThre, lsd_denThre, pseBin = 0.3, 0.6, 22.5, 0.7, 1024
z_occ_max_dis = 1.0   # LSD/baseFunc.h:60
rdp_leastPoint, rdp_threLine, rdp_leastDist = 3, 0.08, 0.5   # RDP defaults, LSD/baseFunc.h:70-72


class lsd_params(C.Structure):
    _fields_ = [("sca", C.c_double), ("sig", C.c_double), ("angThre", C.c_double), ("denThre", C.c_double),
                ("pseBin", C.c_int)]


class lsd_line(C.Structure):  # == structLinesInfo, LSD/baseFunc.h:33-44
    _fields_ = [(n, C.c_double) for n in ("k", "b", "dx", "dy", "x1", "y1", "x2", "y2", "len")] + [("orient", C.c_int)]


# numpy view of structLinesInfo (80 bytes incl. tail padding)
LINE_DTYPE = np.dtype([("k", "f8"), ("b", "f8"), ("dx", "f8"), ("dy", "f8"), ("x1", "f8"), ("y1", "f8"),
                       ("x2", "f8"), ("y2", "f8"), ("len", "f8"), ("orient", "i4"), ("_pad", "i4")])
SEED_DTYPE = np.dtype([("order_idx", "i4"), ("x", "i4"), ("y", "i4"), ("num", "i4"), ("outcome", "i4"),
                       ("final_num", "i4"), ("logNFA", "f8")])
assert LINE_DTYPE.itemsize == 80 == C.sizeof(lsd_line)


class lsd_map_param(C.Structure):      # structMapParam, LSD/baseFunc.h:25-31
    _fields_ = [("oriMapCol", C.c_int), ("oriMapRow", C.c_int), ("mapResol", C.c_double), ("mapOriX", C.c_double), ("mapOriY", C.c_double)]


class lsd_position(C.Structure):  # == structPosition, LSD/baseFunc.h:46-50
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("ang", C.c_double)]


POS_DTYPE = np.dtype([("x", "f8"), ("y", "f8"), ("ang", "f8")])
SCORE_DTYPE = np.dtype([("x", "f8"), ("y", "f8"), ("ang", "f8"), ("score", "f8")])   # lsd_match_score


# lsd_comm (include/lsd_hip.h): rank, world, an all-gather callback of device buffers on a stream, and its user pointer
ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class lsd_comm(C.Structure):
    _fields_ = [("rank", C.c_int), ("world", C.c_int), ("all_gather", ALL_GATHER_FN), ("user", C.c_void_p)]


class LsdError(RuntimeError):
    """A non-zero status of the C ABI.  `partial`: where the C side says its outputs are valid nevertheless (LSD_ERR_CAPACITY /
    LSD_ERR_INTERNAL of the gather, LSD_ERR_CAPACITY of FeatureScan), what the call would have returned."""
    def __init__(self, status, msg, partial=None):
        super().__init__("lsd_hip status %d: %s" % (status, msg))
        self.status = status
        self.partial = partial


_lib = None


def load_library(path=None):
    """Loads liblsdhip.so (built by __graft_entry__.build() / make -C csrc).  Raises if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("LSD_HIP_LIB") or LIB_PATH      # LSD_HIP_LIB: developer A/B builds
    try:
        # torch bundles its own libamdhip64.so.7; when torch is used in the same process (device buffers,
        # streams, RCCL) it must be the HIP runtime that gets loaded first, or the two runtimes clash.
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(p):
        raise ImportError("liblsdhip.so is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "-- there is no CPU fallback" % p)
    L = C.CDLL(p)
    vp, i, sz, dbl = C.c_void_p, C.c_int, C.c_size_t, C.c_double
    L.lsd_create.restype = i; L.lsd_create.argtypes = [C.POINTER(vp), i]
    L.lsd_destroy.restype = None; L.lsd_destroy.argtypes = [vp]
    L.lsd_strerror.restype = C.c_char_p; L.lsd_strerror.argtypes = [i]
    L.lsd_last_error.restype = C.c_char_p; L.lsd_last_error.argtypes = [vp]
    L.lsd_default_params.restype = None; L.lsd_default_params.argtypes = [C.POINTER(lsd_params)]
    L.lsd_abi_version.restype = i; L.lsd_abi_version.argtypes = []
    L.lsd_free.restype = None; L.lsd_free.argtypes = [vp]
    L.lsd_run.restype = i
    L.lsd_run.argtypes = [vp, vp, i, i, sz, C.POINTER(lsd_params), vp, sz, C.POINTER(vp), C.POINTER(i)]
    L.lsd_run_batch.restype = i
    L.lsd_run_batch.argtypes = [vp, vp, i, i, i, C.POINTER(lsd_params), vp, C.POINTER(vp), C.POINTER(i)]
    L.lsd_enqueue_batch_device.restype = i
    L.lsd_enqueue_batch_device.argtypes = [vp, vp, i, i, i, C.POINTER(lsd_params), C.c_uint, vp, vp, i, vp, vp]
    L.lsd_reserve.restype = i; L.lsd_reserve.argtypes = [vp, i, i, i]
    L.lsd_synchronize.restype = i; L.lsd_synchronize.argtypes = [vp]
    L.lsd_scaled_size.restype = None; L.lsd_scaled_size.argtypes = [i, i, dbl, C.POINTER(i), C.POINTER(i)]
    L.lsd_set_stop_after.restype = i; L.lsd_set_stop_after.argtypes = [vp, i]
    L.lsd_set_trace.restype = i; L.lsd_set_trace.argtypes = [vp, i]
    L.lsd_set_region_waves.restype = i; L.lsd_set_region_waves.argtypes = [vp, i]
    L.lsd_set_region_help.restype = i; L.lsd_set_region_help.argtypes = [vp, i]
    L.lsd_debug_set_stamp_budget.restype = i; L.lsd_debug_set_stamp_budget.argtypes = [vp, C.c_uint]
    if hasattr(L, "lsd_set_cost_history") or not os.environ.get("LSD_HIP_LIB"):
        L.lsd_set_cost_history.restype = i; L.lsd_set_cost_history.argtypes = [vp, i]
    if hasattr(L, "lsd_debug_set_tuning") or not os.environ.get("LSD_HIP_LIB"):
        L.lsd_debug_set_tuning.restype = i; L.lsd_debug_set_tuning.argtypes = [vp, C.c_char_p, i]
    L.lsd_set_host_max_lines.restype = i; L.lsd_set_host_max_lines.argtypes = [vp, i]
    L.lsd_debug_fetch.restype = i; L.lsd_debug_fetch.argtypes = [vp, i, i, vp, sz]
    L.lsd_last_timings.restype = i; L.lsd_last_timings.argtypes = [vp, C.POINTER(C.c_float)]
    L.lsd_map_cache.restype = i; L.lsd_map_cache.argtypes = [vp, vp, i, i, sz, dbl, dbl, vp]
    L.lsd_enqueue_map_cache_device.restype = i
    L.lsd_enqueue_map_cache_device.argtypes = [vp, vp, i, i, i, dbl, dbl, vp, vp]
    L.lsd_occupancy_to_map.restype = i; L.lsd_occupancy_to_map.argtypes = [vp, vp, i, i, vp, sz]
    L.lsd_enqueue_occupancy_to_map_device.restype = i
    L.lsd_enqueue_occupancy_to_map_device.argtypes = [vp, vp, sz, vp, vp]
    L.lsd_scan_to_map_match.restype = i
    L.lsd_scan_to_map_match.argtypes = [vp, vp, i, i, vp, i, vp, i, vp, i, lsd_position, lsd_position, vp, i, dbl, dbl, vp]
    L.lsd_enqueue_scan_to_map_match_device.restype = i
    L.lsd_enqueue_scan_to_map_match_device.argtypes = [vp, vp, i, i, vp, vp, vp, i, lsd_position, lsd_position, vp, i, dbl, dbl, vp, vp]
    if hasattr(L, "lsd_feature_scan_batch") or not os.environ.get("LSD_HIP_LIB"):      # (a developer A/B build may predate this entry point)
        L.lsd_feature_scan_batch.restype = i
        L.lsd_feature_scan_batch.argtypes = [vp, vp, vp, i, i, lsd_map_param, i, dbl, dbl, vp, vp, vp, i, vp, vp, vp]
        L.lsd_enqueue_feature_scan_batch_device.restype = i
        L.lsd_enqueue_feature_scan_batch_device.argtypes = [vp, vp, vp, i, i, lsd_map_param, i, dbl, dbl, vp, vp, vp, i, vp, vp, vp, vp]
    if hasattr(L, "lsd_gather_lines") or not os.environ.get("LSD_HIP_LIB"):
        L.lsd_shard_range.restype = None; L.lsd_shard_range.argtypes = [i, i, i, C.POINTER(i), C.POINTER(i)]
        L.lsd_gather_layout.restype = i; L.lsd_gather_layout.argtypes = [i, i, C.POINTER(i), C.POINTER(sz)]
        L.lsd_comm_from_rccl.restype = i; L.lsd_comm_from_rccl.argtypes = [vp, C.POINTER(lsd_comm)]
        L.lsd_gather_lines.restype = i; L.lsd_gather_lines.argtypes = [vp, C.POINTER(lsd_comm), vp, vp, i, i, i, i, vp, vp, vp]
        L.lsd_gather_unpack.restype = i; L.lsd_gather_unpack.argtypes = [vp, vp, i, i, i, vp, vp, sz]
    if hasattr(L, "lsd_shard_balanced") or not os.environ.get("LSD_HIP_LIB"):      # (an A/B build with the gather entry points may predate these two)
        L.lsd_shard_balanced.restype = i; L.lsd_shard_balanced.argtypes = [vp, i, i, vp]
        L.lsd_last_region_cycles.restype = i; L.lsd_last_region_cycles.argtypes = [vp, i, vp]
    if hasattr(L, "lsd_last_sensitivity") or not os.environ.get("LSD_HIP_LIB"):
        L.lsd_last_sensitivity.restype = i; L.lsd_last_sensitivity.argtypes = [vp, i, vp]
    L.lsd_debug_calibrate.restype = i; L.lsd_debug_calibrate.argtypes = [vp, sz]
    L.lsd_debug_eval_math.restype = i; L.lsd_debug_eval_math.argtypes = [vp, i, vp, vp, vp, vp, sz]
    if path is None:
        _lib = L
    return L


EXPORTED_SYMBOLS = ["lsd_create", "lsd_destroy", "lsd_strerror", "lsd_last_error", "lsd_default_params",
                    "lsd_abi_version", "lsd_free", "lsd_run", "lsd_run_batch", "lsd_enqueue_batch_device",
                    "lsd_reserve", "lsd_synchronize", "lsd_scaled_size", "lsd_set_stop_after", "lsd_set_trace", "lsd_set_region_waves", "lsd_set_region_help", "lsd_debug_set_stamp_budget", "lsd_debug_set_tuning", "lsd_set_cost_history", "lsd_shard_balanced", "lsd_last_region_cycles", "lsd_last_sensitivity", "lsd_set_host_max_lines",
                    "lsd_debug_fetch", "lsd_last_timings", "lsd_debug_eval_math", "lsd_debug_calibrate", "lsd_map_cache",
                    "lsd_enqueue_map_cache_device", "lsd_occupancy_to_map", "lsd_enqueue_occupancy_to_map_device",
                    "lsd_scan_to_map_match", "lsd_enqueue_scan_to_map_match_device",
                    "lsd_feature_scan_batch", "lsd_enqueue_feature_scan_batch_device",
                    "lsd_shard_range", "lsd_gather_layout", "lsd_comm_from_rccl", "lsd_gather_lines", "lsd_gather_unpack"]


def shard_range(n_items, world, rank):
    """lsd_shard_range: the contiguous shard [lo, hi) of n_items images that `rank` of `world` takes."""
    lo, hi = C.c_int(), C.c_int()
    load_library().lsd_shard_range(n_items, world, rank, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def shard_balanced(costs, world):
    """lsd_shard_balanced -> perm int32 [n]: rank r takes images perm[lo_r:hi_r] (shard_range) and every rank carries about the same cost."""
    c = np.ascontiguousarray(costs, np.int64)
    perm = np.zeros(len(c), np.int32)
    st = load_library().lsd_shard_balanced(c.ctypes.data, len(c), world, perm.ctypes.data)
    if st != LSD_OK:
        raise LsdError(st, load_library().lsd_strerror(st).decode())
    return perm


def gather_layout(n_total, world):
    """lsd_gather_layout -> (images of the largest shard, int32 words of the gathered counts array)."""
    per, words = C.c_int(), C.c_size_t()
    st = load_library().lsd_gather_layout(n_total, world, C.byref(per), C.byref(words))
    if st != LSD_OK:
        raise LsdError(st, load_library().lsd_strerror(st).decode())
    return per.value, words.value


def gather_unpack(counts_all, slabs_all, n_total, world, cap_rows):
    """lsd_gather_unpack on HOST copies of the gathered arrays (int32 [world, per + 2], LINE_DTYPE / 80-byte records [world, cap_rows]):
    returns (offsets int32 [n_total + 1], lines LINE_DTYPE in global image order); raises LsdError(LSD_ERR_CAPACITY) if a rank dropped rows
    and LsdError(LSD_ERR_INTERNAL) if the region stage gave an image up -- the exception's `partial` holds what did arrive."""
    L = load_library()
    ca = np.ascontiguousarray(counts_all, np.int32)
    sl = np.ascontiguousarray(slabs_all).view(np.uint8).reshape(-1, 80)
    offs = np.zeros(n_total + 1, np.int32)
    per, _ = gather_layout(n_total, world)
    total = int(ca.reshape(world, per + 2)[:, :per].sum())
    lines = np.zeros(max(total, 1), LINE_DTYPE)
    st = L.lsd_gather_unpack(ca.ctypes.data, sl.ctypes.data, n_total, world, cap_rows, offs.ctypes.data, lines.ctypes.data, len(lines))
    if st in (LSD_ERR_CAPACITY, LSD_ERR_INTERNAL):
        raise LsdError(st, L.lsd_strerror(st).decode(), partial=(offs, lines[:offs[-1]]))
    if st != LSD_OK:
        raise LsdError(st, L.lsd_strerror(st).decode())
    return offs, lines[:offs[-1]]


def make_params(sca=lsd_sca, sig=lsd_sig, angThre=lsd_angThre, denThre=lsd_denThre, pseBin=pseBin):
    return lsd_params(float(sca), float(sig), float(angThre), float(denThre), int(pseBin))


def scaled_size(cols, rows, sca=lsd_sca):
    w, h = C.c_int(), C.c_int()
    load_library().lsd_scaled_size(cols, rows, sca, C.byref(w), C.byref(h))
    return w.value, h.value


class Context:
    """One lsd_ctx: one GPU, one stream, one HBM workspace (one per process rank)."""

    def __init__(self, device=0):
        self.L = load_library()
        h = C.c_void_p()
        st = self.L.lsd_create(C.byref(h), int(device))
        if st != LSD_OK:
            raise LsdError(st, self.L.lsd_strerror(st).decode())
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.L.lsd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, st, allow=()):
        if st != LSD_OK and st not in allow:
            raise LsdError(st, self.L.lsd_strerror(st).decode() + " / " + self.L.lsd_last_error(self.h).decode())
        return st

    # -- host-buffer entry points -----------------------------------------------------------------
    def run(self, map_u8, params=None, want_lineim=True):
        """lsd_run on a C-contiguous uint8 image; the image is rewritten in place like the reference does."""
        assert map_u8.dtype == np.uint8 and map_u8.ndim == 2
        rows, cols = map_u8.shape
        p = params or make_params()
        line_im = np.zeros((rows, cols), np.uint8) if want_lineim else None
        lines_p, n = C.c_void_p(), C.c_int()
        st = self.L.lsd_run(self.h, map_u8.ctypes.data, cols, rows, map_u8.strides[0], C.byref(p),
                            line_im.ctypes.data if want_lineim else None, cols, C.byref(lines_p), C.byref(n))
        lines = np.zeros(n.value if lines_p else 0, LINE_DTYPE)
        if len(lines):
            C.memmove(lines.ctypes.data, lines_p, 80 * len(lines))
            lines["_pad"] = 0
        self.L.lsd_free(lines_p)                                   # (before any raise: the C side has handed the buffer over)
        self._chk(st)
        return lines, line_im

    def run_batch(self, maps_u8, params=None, want_lineim=True):
        """lsd_run_batch on an [n, rows, cols] uint8 array (rewritten in place)."""
        assert maps_u8.dtype == np.uint8 and maps_u8.ndim == 3 and maps_u8.flags.c_contiguous
        n, rows, cols = maps_u8.shape
        p = params or make_params()
        line_ims = np.zeros((n, rows, cols), np.uint8) if want_lineim else None
        lines_p = C.c_void_p()
        offs = (C.c_int * (n + 1))()
        st = self.L.lsd_run_batch(self.h, maps_u8.ctypes.data, n, cols, rows, C.byref(p),
                                  line_ims.ctypes.data if want_lineim else None, C.byref(lines_p), offs)
        offsets = np.frombuffer(offs, np.int32).copy()
        lines = np.zeros(int(offsets[-1]) if lines_p else 0, LINE_DTYPE)
        if len(lines):
            C.memmove(lines.ctypes.data, lines_p, 80 * len(lines))
            lines["_pad"] = 0
        self.L.lsd_free(lines_p)                                   # (before any raise: the C side has handed the buffer over)
        self._chk(st)
        return lines, offsets, line_ims

    # -- device-resident batch --------------------------------------------------------------------
    def enqueue_device(self, d_maps, n, cols, rows, d_lines, max_lines, d_counts, d_line_ims=None, params=None,
                       flags=0, stream=None):
        """Pointers are raw device addresses (e.g. torch tensor .data_ptr()); asynchronous."""
        p = params or make_params()
        return self._chk(self.L.lsd_enqueue_batch_device(self.h, d_maps, n, cols, rows, C.byref(p), flags, d_line_ims,
                                                         d_lines, max_lines, d_counts, stream))

    def gather_lines(self, comm, d_lines, d_counts, n_local, max_lines, n_total, cap_rows, d_counts_all, d_slabs_all, stream=None):
        """lsd_gather_lines: comm is an lsd_comm (dist.torch_comm / lsd_comm_from_rccl); pointers are raw device addresses."""
        return self._chk(self.L.lsd_gather_lines(self.h, C.byref(comm), d_lines, d_counts, n_local, max_lines, n_total, cap_rows,
                                                 d_counts_all, d_slabs_all, stream))

    def map_cache(self, map_u8, res, z_occ_max_dis=1.0):
        """lsd_map_cache on a uint8 image (read-only); returns float64 [rows, cols]."""
        assert map_u8.dtype == np.uint8 and map_u8.ndim == 2
        rows, cols = map_u8.shape
        out = np.zeros((rows, cols), np.float64)
        self._chk(self.L.lsd_map_cache(self.h, map_u8.ctypes.data, cols, rows, map_u8.strides[0], float(res),
                                       float(z_occ_max_dis), out.ctypes.data))
        return out

    def enqueue_map_cache_device(self, d_maps, n, cols, rows, res, z_occ_max_dis, d_out, stream=None):
        return self._chk(self.L.lsd_enqueue_map_cache_device(self.h, d_maps, n, cols, rows, float(res),
                                                             float(z_occ_max_dis), d_out, stream))

    def scan_to_map_match(self, map_cache, map_lines, scan_lines, scan_im_points, lidar_pose, last_pose, pairs,
                          z_occ=1.0, max_esti_dist=60.0):
        """lsd_scan_to_map_match: map_cache float64 [rows, cols]; lines LINE_DTYPE arrays; scan_im_points POS_DTYPE (or
        [n, 3] float64); poses (x, y[, ang]); pairs int32 [m, 2] = (cntMapLine, cntScanLine).  Returns SCORE_DTYPE [m, 4]."""
        mc = np.ascontiguousarray(map_cache, np.float64)
        rows, cols = mc.shape
        ml = np.ascontiguousarray(map_lines, LINE_DTYPE); sl = np.ascontiguousarray(scan_lines, LINE_DTYPE)
        pts = np.ascontiguousarray(scan_im_points).view(np.float64).reshape(-1, 3)
        pr = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        out = np.zeros((len(pr), 4), SCORE_DTYPE)
        mk = lambda p: lsd_position(float(p[0]), float(p[1]), float(p[2]) if len(p) > 2 else 0.0)
        self._chk(self.L.lsd_scan_to_map_match(self.h, mc.ctypes.data, cols, rows, ml.ctypes.data, len(ml), sl.ctypes.data, len(sl),
                                               pts.ctypes.data, len(pts), mk(lidar_pose), mk(last_pose), pr.ctypes.data, len(pr),
                                               float(z_occ), float(max_esti_dist), out.ctypes.data))
        return out

    def feature_scan_batch(self, scans, lens, map_param, region_point_limit=rdp_leastPoint, thre_line=rdp_threLine,
                           line_dist_thre_m=rdp_leastDist, pts_cap=4096):
        """lsd_feature_scan_batch: scans float64 [n, stride, 2] = (range, angle), lens int32 [n] finite readings per scan,
        map_param = (oriMapCol, oriMapRow, mapResol, mapOriX, mapOriY).  Returns a list of dicts like FeatureScan() below."""
        sc = np.ascontiguousarray(scans, np.float64)
        n, stride = sc.shape[0], sc.shape[1]
        ln = np.ascontiguousarray(lens, np.int32)
        lines = np.zeros((n, 360), LINE_DTYPE); pts = np.zeros((n, pts_cap, 3), np.float64)
        nl = np.zeros(n, np.int32); npt = np.zeros(n, np.int32); lp = np.zeros((n, 2), np.float64); sz = np.zeros((n, 2), np.int32)
        mp = lsd_map_param(int(map_param[0]), int(map_param[1]), float(map_param[2]), float(map_param[3]), float(map_param[4]))
        # (more than 360 line records in a scan -- the reference would overrun its array there -- raises LsdError(LSD_ERR_CAPACITY), with
        #  the stored records, which the C side says are valid, in the exception's `partial`)
        st = self.L.lsd_feature_scan_batch(self.h, sc.ctypes.data, ln.ctypes.data, n, stride, mp, int(region_point_limit), float(thre_line),
                                           float(line_dist_thre_m), lines.ctypes.data, nl.ctypes.data, pts.ctypes.data, pts_cap,
                                           npt.ctypes.data, lp.ctypes.data, sz.ctypes.data)
        if st != LSD_ERR_CAPACITY:
            self._chk(st)
        out = []
        for i in range(n):
            nl[i] = min(int(nl[i]), 360)
            if npt[i] > pts_cap:
                raise RuntimeError("scan %d marks %d pixels, more than pts_cap" % (i, npt[i]))
            p = pts[i, :npt[i]].copy()
            im = np.zeros((max(int(sz[i, 1]), 0), max(int(sz[i, 0]), 0)), np.uint8)        # FS.lineIm (myRDP.cpp:38, :141)
            if len(p):
                im[p[:, 1].astype(int), p[:, 0].astype(int)] = 255
            out.append(dict(linesInfo=lines[i, :nl[i]].copy(), len_linesInfo=int(nl[i]), scanImPoint=p, lidarPos=(float(lp[i, 0]), float(lp[i, 1])),
                            lineIm=im))
        if st == LSD_ERR_CAPACITY:
            raise LsdError(st, self.L.lsd_strerror(st).decode(), partial=out)
        return out

    def occupancy_to_map(self, grid_i8):
        """lsd_occupancy_to_map on an int8 [rows, cols] OccupancyGrid; returns the uint8 map."""
        assert grid_i8.dtype == np.int8 and grid_i8.ndim == 2 and grid_i8.flags.c_contiguous
        rows, cols = grid_i8.shape
        out = np.zeros((rows, cols), np.uint8)
        self._chk(self.L.lsd_occupancy_to_map(self.h, grid_i8.ctypes.data, cols, rows, out.ctypes.data, out.strides[0]))
        return out

    def enqueue_occupancy_to_map_device(self, d_grid, n_cells, d_map, stream=None):
        return self._chk(self.L.lsd_enqueue_occupancy_to_map_device(self.h, d_grid, n_cells, d_map, stream))

    def reserve(self, n, cols, rows):
        self._chk(self.L.lsd_reserve(self.h, n, cols, rows))

    def synchronize(self):
        self._chk(self.L.lsd_synchronize(self.h))

    def fetch_stats_block(self, n):
        """The raw counter records (the values of fetch(i, DBG_STATS, ...), in its order) of the first n images of the last
        batch as an (n, 48) int64 array, in one copy (developer probes)."""
        a = np.zeros((n, 48), np.int64)
        self._chk(self.L.lsd_debug_fetch(self.h, 0, DBG_STATS, a.ctypes.data, a.nbytes))
        return a

    def last_region_cycles(self, n):
        """lsd_last_region_cycles: shader clocks of the region stage per image of the last batch (the cost shard_balanced deals by)."""
        a = np.zeros(n, np.int64)
        self._chk(self.L.lsd_last_region_cycles(self.h, n, a.ctypes.data))
        return a

    def last_sensitivity(self, n):
        """lsd_last_sensitivity: per image of the last batch, the number of decisions taken within the noise of the reference's libm
        (0: the image's result is the reference's under any libm within one ulp of correct rounding)."""
        a = np.zeros(n, np.int32)
        self._chk(self.L.lsd_last_sensitivity(self.h, n, a.ctypes.data))
        return a

    def timings(self):
        ms = (C.c_float * 6)()
        self._chk(self.L.lsd_last_timings(self.h, ms))
        return dict(zip(("gauss", "gradient", "sort", "region", "lines", "total"), [float(x) for x in ms]))

    # -- introspection ----------------------------------------------------------------------------
    def set_stop_after(self, stage):
        self._chk(self.L.lsd_set_stop_after(self.h, stage))

    def set_trace(self, on):
        self._chk(self.L.lsd_set_trace(self.h, 1 if on else 0))

    def set_host_max_lines(self, max_lines):
        """Line capacity per image of run / run_batch (default 8192); more lines -> LsdError(LSD_ERR_CAPACITY)."""
        self._chk(self.L.lsd_set_host_max_lines(self.h, max_lines))

    def set_cost_history(self, on):
        """lsd_set_cost_history: start the images of a batch in the order of their cost in this context's previous call (same maps from call to call)."""
        self._chk(self.L.lsd_set_cost_history(self.h, 1 if on else 0))

    def debug_set_tuning(self, name, value):
        """Test / developer hook (lsd_debug_set_tuning): a schedule setting of the region stage by name; no result depends on any."""
        self._chk(self.L.lsd_debug_set_tuning(self.h, name.encode(), int(value)))

    def debug_set_stamp_budget(self, grows):
        """Test hook (lsd_debug_set_stamp_budget): curMap stamp ids per wavefront and run before the stamps are cleared."""
        self._chk(self.L.lsd_debug_set_stamp_budget(self.h, grows))

    def set_region_waves(self, waves):
        """0: automatic, 4 / 8: force the region-stage variant (results are identical)."""
        self._chk(self.L.lsd_set_region_waves(self.h, waves))

    def set_region_help(self, waves):
        """Helper wavefronts per image of the region stage's help across workgroups (lsd_set_region_help): 0 switches it off (the
        setting for several batches in flight on several contexts), -1 restores the default.  Results are identical."""
        self._chk(self.L.lsd_set_region_help(self.h, waves))

    def fetch(self, image, what, shape_wh):
        """Returns the intermediate `what` (DBG_*) of image `image` of the last run as a numpy array."""
        w, h = shape_wh
        npx = w * h

        def get(kind, dtype, count):
            a = np.zeros(max(count, 1), dtype)
            self._chk(self.L.lsd_debug_fetch(self.h, image, kind, a.ctypes.data, a.nbytes))
            return a[:count]

        if what in (DBG_GAUSS, DBG_MAG, DBG_DEG):
            return get(what, np.float64, npx).reshape(h, w)
        if what == DBG_STATE:
            return get(what, np.uint32, npx).reshape(h, w)
        if what == DBG_NB:
            return int(get(what, np.int32, 1)[0])
        if what == DBG_NSEED:
            return int(get(what, np.int32, 1)[0])
        if what == DBG_MAXGRAD:
            return float(get(what, np.float64, 1)[0])
        if what == DBG_ORDER:
            return get(what, np.uint32, npx)[:self.fetch(image, DBG_NB, shape_wh)]
        if what == DBG_ORDER_VAL:
            return get(what, np.uint16, npx)[:self.fetch(image, DBG_NB, shape_wh)]
        if what == DBG_STATS:
            v = get(what, np.int64, 48)
            d = dict(zip(("grow_calls", "grown_px", "nfa_calls", "rrr_calls", "rrr_passes", "rrr_sentinel_drops",
                          "rrr_oob_reads", "cycles_refill", "cycles_total", "cycles_grow", "cycles_rect",
                          "cycles_nfa", "cycles_mark", "small_bails", "wait_noslot", "seeds", "exact_angle_evals",
                          "tile_fetches", "batches", "cycles_tiles", "spec_redos", "spec_discards", "cycles_wait", "small_steps",
                          "refill_rounds", "cycles_eval", "cycles_sums", "cycles_refine", "cycles_small", "cycles_select", "cycles_commit",
                          "wait_noseed", "requeued_ahead", "cycles_eval_at_cursor", "depth_end", "nfa_min_abs_enc", "nfa_min_gap_enc", "help_exports", "help_evals", "near_ties",
                          "wd_commit", "wd_next", "wd_nseeds", "wd_state", "wd_nbig", "wd_lock", "wd_pend", "wd_wave"),
                         [int(x) for x in v]))
            d["set_answers"], d["sets_founded"] = int(v[41]), int(v[42])   # evaluations answered by a certified uniform set / sets founded (k_region.hip)
            d["cycles_nfa_count"], d["nfa_tail_iters"] = int(v[43]), int(v[44])   # developer build: cycles of the NFA's pixel count, iterations of its tail sum
            d["nfa_bracket_misses"] = int(v[40])   # stopping tests of the NFA's tail left to the correctly rounded pow / log10 (an image the watchdog gave up keeps its record here instead)
            # how close RectangleImprover's comparisons came to a tie, as margins (distance of the operands over what an ulp of exp / log10 /
            # pow can move them; k_region.hip: improve()): smallest for a logNFA compared with 0, smallest for two compared NFA values (inf: none seen)
            for k in ("nfa_min_abs", "nfa_min_gap"):
                enc = d.pop(k + "_enc")
                d[k] = float("inf") if enc == 0 else float(np.array([0x7ff0000000000000 - enc], np.uint64).view(np.float64)[0])
            return d
        if what == DBG_SEEDS:
            ns = self.fetch(image, DBG_NSEED, shape_wh)
            return get(what, SEED_DTYPE, ns)
        raise ValueError(what)

    def eval_math(self, fn, a, b=None):
        """Device sin/cos (fn 0), atan2(a, b) (fn 1), atan (fn 2) of float64 arrays, or the region stage's fp32 estimate of
        sin/cos of the packed angle (fn 3) (test hook)."""
        a = np.ascontiguousarray(a, np.float64)
        b = None if b is None else np.ascontiguousarray(b, np.float64)
        o0, o1 = np.zeros_like(a), np.zeros_like(a)
        self._chk(self.L.lsd_debug_eval_math(self.h, fn, a.ctypes.data, None if b is None else b.ctypes.data,
                                             o0.ctypes.data, o1.ctypes.data, a.size))
        return o0, o1

    def fetch_recs(self, image, count):
        a = np.zeros(max(count * 12, 1), np.float64)
        self._chk(self.L.lsd_debug_fetch(self.h, image, DBG_RECS, a.ctypes.data, a.nbytes))
        return a[:count * 12].reshape(count, 12)


# ---- the reference's own names (LSD/myLSD.h:123-132) ----------------------------------------------
class structLSD:
    """structLSD (LSD/myLSD.h:123-127): lineIm (CV_8UC1 rows x cols, 0/255), linesInfo, len_linesInfo."""

    def __init__(self, lineIm, linesInfo):
        self.lineIm = lineIm
        self.linesInfo = linesInfo
        self.len_linesInfo = len(linesInfo)


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


def myLineSegmentDetector(MapGray, oriMapCol, oriMapRow, sca, sig, angThre, denThre, pseBin, ctx=None):
    """mylsd::myLineSegmentDetector (LSD/myLSD.h:132, LSD/myLSD.cpp:129): same argument meaning, same
    in-place rewrite of MapGray, returns structLSD.  No validation beyond the C ABI's, like the reference."""
    if MapGray.shape != (oriMapRow, oriMapCol):
        raise LsdError(LSD_ERR_INVALID, "MapGray must be oriMapRow x oriMapCol")
    ctx = ctx or default_context()
    lines, line_im = ctx.run(MapGray, make_params(sca, sig, angThre, denThre, pseBin))
    return structLSD(line_im, lines)


def createMapCache(MapGray, res, ctx=None):
    """mylsd::createMapCache (LSD/myLSD.h:131, LSD/myLSD.cpp:11): CV_64FC1-like float64 array, metres, capped at
    z_occ_max_dis.  Call it before myLineSegmentDetector, which rewrites MapGray."""
    return (ctx or default_context()).map_cache(MapGray, res, z_occ_max_dis)


# FeatureAssociation constants, LSD/baseFunc.h:80-86
ignoreScanLength, scanToMapDiff, maxEstiDist = 40, 0.35, 60


def match_pairs(map_lines, scan_lines):
    """The (cntMapLine, cntScanLine) pairs FeatureAssociation hands to its thread pool (LSD/myFA.cpp:28-58), in its order."""
    pairs = []
    for cs in range(len(scan_lines)):
        ls = float(scan_lines["len"][cs])
        if ls < ignoreScanLength:
            continue
        ld = ls * scanToMapDiff
        for cm in range(len(map_lines)):
            lm = float(map_lines["len"][cm])
            if lm < ls - ld or lm > ls + ld:
                continue
            pairs.append((cm, cs))
    return np.array(pairs, np.int32).reshape(-1, 2)


def FeatureScan(mapParam, lidarPointPolar, RegionPointLimitNumber=rdp_leastPoint, threLine=rdp_threLine, lineDistThreM=rdp_leastDist, ctx=None):
    """myrdp::FeatureScan (LSD/myRDP.cpp:9) for one scan: mapParam = (oriMapCol, oriMapRow, mapResol, mapOriX, mapOriY), lidarPointPolar float64
    [len_lp, 2] = (range, angle) of the finite readings (the caller's read loop drops the infinite ones, LSD/main_on_windows.cpp:115-121).
    Returns dict(linesInfo, len_linesInfo, scanImPoint [m, 3], lidarPos (x, y), lineIm) -- structFeatureScan (LSD/myRDP.h:55-61)."""
    own = ctx is None
    ctx = ctx or Context(0)
    try:
        sc = np.ascontiguousarray(lidarPointPolar, np.float64).reshape(1, -1, 2)
        return ctx.feature_scan_batch(sc, [sc.shape[1]], mapParam, RegionPointLimitNumber, threLine, lineDistThreM)[0]
    finally:
        if own:
            ctx.close()


def ScanToMapMatch(mapCache, mapLinesInfo, scanLinesInfo, scanImPoint, lidarPose, lastPose, ctx=None):
    """The matching stage of myfa::FeatureAssociation (LSD/myFA.cpp:28-100): every admissible (scan line, map line) pair x 4
    matchings scored on the device, candidates with score < 3 kept (:262) and sorted by score (:98, CompScore :398-402).
    Returns SCORE_DTYPE records (the reference's structScore without the debug pointer)."""
    pairs = match_pairs(mapLinesInfo, scanLinesInfo)
    if len(pairs) == 0:
        return np.zeros(0, SCORE_DTYPE)
    sc = (ctx or default_context()).scan_to_map_match(mapCache, mapLinesInfo, scanLinesInfo, scanImPoint, lidarPose, lastPose, pairs,
                                                      z_occ_max_dis, maxEstiDist).ravel()
    keep = sc[sc["score"] < 3]
    return keep[np.argsort(keep["score"], kind="stable")]


def mapCallback(data, oriMapCol, oriMapRow, mapResol, ctx=None):
    """The ROS node's map callback (LSD/main_on_linux.cpp:97-135): OccupancyGrid cells -> mapValue -> mapCache (with
    the callback's z_occ_max_dis = 2, :126-127) and the LSD result.  Returns (mapValue as rewritten by the LSD call,
    mapCache, structLSD)."""
    cx = ctx or default_context()
    grid = np.ascontiguousarray(np.asarray(data, np.int8).reshape(oriMapRow, oriMapCol))
    mapValue = cx.occupancy_to_map(grid)
    mapCache = cx.map_cache(mapValue, mapResol, 2.0)
    LSD = myLineSegmentDetector(mapValue, oriMapCol, oriMapRow, lsd_sca, lsd_sig, lsd_angThre, lsd_denThre, pseBin, ctx=cx)
    return mapValue, mapCache, LSD


def runLSD(MapGray, oriMapCol=None, oriMapRow=None, sca=lsd_sca, sig=lsd_sig, angThre=lsd_angThre,
           denThre=lsd_denThre, pseBin=pseBin, ctx=None):
    """`runLSD` is the name BASELINE.json's north_star uses; the reference has no such symbol
    (SURVEY section 0.1).  It is an alias of myLineSegmentDetector with the baseFunc.h defaults."""
    rows, cols = MapGray.shape
    return myLineSegmentDetector(MapGray, oriMapCol or cols, oriMapRow or rows, sca, sig, angThre, denThre, pseBin, ctx)
