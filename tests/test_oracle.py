"""Pins the CPU oracle (oracle/lsd_oracle.c) to the reference.

The reference cannot be rebuilt in this image (OpenCV/Eigen headers absent), so the pins are
the reference outputs recorded in SURVEY.md 8c / Appendix A (tests/golden/known_answers.json)
plus the reference's own MATLAB-era golden files for data/mapValue.txt (loose).
"""
import json
import os

import numpy as np
import pytest

from conftest import tile2048

FIXTURES = ["map1", "mapValue", "aisle1", "aisle2", "aisle3", "f3key", "f4key"]


@pytest.mark.parametrize("name", FIXTURES)
def test_counts_match_reference(name, maps, maps_meta, known, oracle):
    nl, lit, csum, ccnt = known["counts"][name]
    m = maps[name].copy()
    mc = oracle.map_cache(m, maps_meta[name]["res"])      # createMapCache runs BEFORE LSD (Q2)
    r = oracle.lsd(m)
    assert len(r["lines"]) == nl
    assert int((r["lineIm"] == 255).sum()) == lit
    assert set(np.unique(r["lineIm"])) <= {0, 255}
    assert abs(float(mc.sum()) - csum) < 1e-5
    assert int((mc < 1.0).sum()) == ccnt


def test_tile2048_matches_reference(maps, known, oracle):
    nl, lit, csum, ccnt = known["counts"]["tile2048"]
    m = tile2048(maps["aisle1"])
    mc = oracle.map_cache(m, 0.025)
    r = oracle.lsd(m)
    assert len(r["lines"]) == nl
    assert int((r["lineIm"] == 255).sum()) == lit
    assert abs(float(mc.sum()) - csum) < 1e-4
    assert int((mc < 1.0).sum()) == ccnt


def test_map1_line_list_bit_exact(maps, known, oracle):
    """Appendix A prints the reference's map1 lines with %.17g (round-trip exact)."""
    r = oracle.lsd(maps["map1"].copy())
    gold = known["map1_lines"]
    assert len(r["lines"]) == len(gold)
    for got, row in zip(r["lines"], gold):
        for f, s in zip(known["map1_lines_fields"], row):
            if f == "orient":
                assert int(got[f]) == int(s)
            else:
                assert float(got[f]) == float(s), (f, got[f], s)


def test_map1_stage_counters(maps, known, oracle):
    c = known["map1_counters"]
    d = oracle.lsd(maps["map1"].copy(), debug=True)["dbg"]
    assert d["nb"] == c["sorted_px"]
    assert d["grow_calls"] == c["grow_calls"]
    assert d["grown_px"] == c["grown_px"]
    assert d["nfa_calls"] == c["nfa_calls"]
    assert (d["rrr_calls"], d["rrr_passes"]) == (c["rrr_calls"], c["rrr_passes"])
    seeds = d["seeds"]
    assert len(seeds) == c["seeds"]
    assert int((seeds["outcome"] == 0).sum()) == c["small_drops"]
    assert int((seeds["outcome"] == 2).sum()) == c["nfa_rejects"]
    assert int(seeds["num"].max()) == c["max_region"]
    assert d["rrr_oob_reads"] == 0          # the reference's UB read (Q6) never fires on fixtures


def test_map2_is_map1(maps):
    assert "map2" not in maps   # make_fixtures.py asserts byte-identity and stores it once


def test_inplace_remap_and_border(maps, oracle):
    """Q2: the caller's image is mutated for y>=1,x>=1 only; row 0 / col 0 keep raw values."""
    src = maps["map1"]
    m = src.copy()
    m[0, 5] = 1; m[7, 0] = 255; m[3, 3] = 1; m[4, 4] = 255
    ref = m.copy()
    oracle.lsd(m, want_lineim=False)
    assert m[0, 5] == 1 and m[7, 0] == 255
    assert m[3, 3] == 255 and m[4, 4] == 0
    inner = ref[1:, 1:]
    exp = np.where(inner == 1, 255, np.where(inner == 255, 0, inner))
    assert np.array_equal(m[1:, 1:], exp)
    assert np.array_equal(m[0, :], ref[0, :]) and np.array_equal(m[:, 0], ref[:, 0])


def test_scaled_border_is_zero(maps, oracle):
    """Q3: row 0 / col 0 of mag, deg, used stay 0."""
    d = oracle.lsd(maps["aisle1"].copy(), debug=True)["dbg"]
    for k in ("mag", "deg", "used0"):
        assert not d[k][0, :].any() and not d[k][:, 0].any()


def test_sort_is_stable_descending(maps, oracle):
    """Q4: descending by bin value, raster order (y-major, then x) among ties."""
    d = oracle.lsd(maps["aisle2"].copy(), debug=True)["dbg"]
    v, x, y = d["ord_v"].astype(np.int64), d["ord_x"].astype(np.int64), d["ord_y"].astype(np.int64)
    assert len(v) == d["nb"] > 0
    assert v.max() <= 1024 and v.min() >= 1
    key = (1024 - v) * (1 << 40) + y * (1 << 20) + x
    assert np.all(np.diff(key) > 0)
    zoom = 1.0 * 1024 / d["maxGrad"]
    vv = np.floor(d["mag"] * zoom).astype(np.int64).clip(max=1024)
    assert int((vv != 0).sum()) == d["nb"]
    assert np.array_equal(vv[y, x], v)


def test_glibc_qsort_comparator_is_stable(oracle):
    """Pins the libc behaviour Q4 relies on: qsort + the reference comparator == stable order."""
    for n, seed in ((10, 1), (1000, 2), (50000, 3)):
        assert oracle.lib().orc_selftest_qsort_stable(n, seed) == 0


def test_used_map_values(maps, oracle):
    d = oracle.lsd(maps["mapValue"].copy(), debug=True)["dbg"]
    assert set(np.unique(d["used"])) <= {0, 1, 2}
    # accepted/rejected regions only ever add marks
    assert np.all((d["used0"] == 1) <= (d["used"] == 1))


# Rows of data/MaplinesInfo.txt that the C++ reference does not reproduce (the file comes from the author's MATLAB prototype,
# SURVEY section 4): measured 3.3 / 8.0 / 6.0 / 23.3 px off.  Every other row is reproduced to <= 0.064 px.
MATLAB_ROWS_NOT_REPRODUCED = (5, 30, 37, 38)
MATLAB_LINE_TOL_PX = 0.1
MATLAB_RASTER_MIN_HITS = 3850          # of 3992 lit golden pixels, after the (+1, +1) shift (measured: 3865)


def matlab_golden_check(lines, line_im, maps):
    """The reference-held golden files for data/mapValue.txt (data/MaplinesInfo.txt, data/MaplineIm.txt), as tightly as
    they support: >= 36 of the 40 golden lines within 0.1 px (endpoints in either order: the prototype lists them swapped),
    with the same length and orientation sign, and >= 3850 of the 3992 lit golden pixels lit exactly, the golden raster
    being 1-based (MATLAB) and therefore shifted by (+1, +1).  Shared with the GPU parity test: no oracle in the loop."""
    gold = maps["matlab_MaplinesInfo"]          # k b dx dy x1 y1 x2 y2 len orient
    P = np.stack([lines["x1"], lines["y1"]], 1); Q = np.stack([lines["x2"], lines["y2"]], 1)
    matched, bad_rows = 0, []
    for gi, g in enumerate(gold):
        a, b = g[4:6], g[6:8]
        e_same = np.maximum(np.abs(P - a).max(1), np.abs(Q - b).max(1))
        e_swap = np.maximum(np.abs(Q - a).max(1), np.abs(P - b).max(1))
        e = np.minimum(e_same, e_swap)
        j = int(np.argmin(e))
        if e[j] <= MATLAB_LINE_TOL_PX:
            matched += 1
            assert abs(lines["len"][j] - g[8]) < 0.05, (gi, lines["len"][j], g[8])
            if np.isfinite(g[0]):
                assert int(lines["orient"][j]) == int(g[9]), gi
                assert abs(abs(lines["dx"][j]) - abs(g[2])) < 2e-3 and abs(abs(lines["dy"][j]) - abs(g[3])) < 2e-3, gi
        else:
            bad_rows.append(gi)
    assert matched >= 36, (matched, bad_rows)
    assert tuple(bad_rows) == MATLAB_ROWS_NOT_REPRODUCED, bad_rows
    lit = maps["matlab_MaplineIm_lit_yx"]
    assert len(lit) == 3992
    ys, xs = lit[:, 0] + 1, lit[:, 1] + 1
    ok = (ys < line_im.shape[0]) & (xs < line_im.shape[1])
    hits = int((line_im[ys[ok], xs[ok]] == 255).sum())
    assert hits >= MATLAB_RASTER_MIN_HITS, hits
    # without the shift the two rasters hardly meet: the shift is a property of the data, not a fudge
    assert int((line_im[lit[:, 0], lit[:, 1]] == 255).sum()) < 100
    return matched, hits


def test_matlab_golden_pins_the_oracle(maps, oracle):
    """The only outputs the reference itself holds for this path; they pin the oracle."""
    r = oracle.lsd(maps["mapValue"].copy())
    matched, hits = matlab_golden_check(r["lines"], r["lineIm"], maps)
    assert (matched, hits) == (36, 3865)


def test_blank_and_tiny_images(oracle):
    r = oracle.lsd(np.zeros((64, 80), np.uint8))
    assert len(r["lines"]) == 0 and not r["lineIm"].any()
    r = oracle.lsd(np.zeros((5, 5), np.uint8))
    assert len(r["lines"]) == 0


def test_synthetic_square_is_deterministic(oracle):
    rng = np.random.default_rng(5)
    m = np.zeros((300, 400), np.uint8)
    m[60:240, 80] = 1; m[60:240, 320] = 1; m[60, 80:321] = 1; m[240, 80:321] = 1
    m[rng.integers(1, 299, 200), rng.integers(1, 399, 200)] = 1
    a = oracle.lsd(m.copy(), debug=True)
    b = oracle.lsd(m.copy(), debug=True)
    assert len(a["lines"]) >= 4
    assert a["lines"].tobytes() == b["lines"].tobytes()
    assert np.array_equal(a["dbg"]["used"], b["dbg"]["used"])


def test_oracle_asan_clean(maps, oracle):
    """CPU-only sanitizer pass (GPU sanitizers are unavailable on the pool)."""
    import subprocess, sys, os
    so = oracle.build(asan=True)
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "from oracle import oracle\n"
        "L = oracle.lib(%r)\n"
        "z = np.load(%r)\n"
        "for k in ('map1', 'aisle1'):\n"
        "    r = oracle.lsd(z[k].copy(), debug=True, _lib=L)\n"
        "    oracle.map_cache(z[k].copy(), 0.05, _lib=L)\n"
        "print('OK')\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), so,
         os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "maps.npz"))
    import glob
    asan = sorted(glob.glob("/usr/lib/gcc/x86_64-linux-gnu/*/libasan.so"))
    if not asan:
        pytest.skip("libasan not installed")
    env = dict(os.environ, LD_PRELOAD=asan[-1], ASAN_OPTIONS="detect_leaks=0")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "OK" in p.stdout, p.stderr[-2000:]


def test_occupancy_to_map_exhaustive(oracle):
    """All 256 cell values through the restated map-callback loop (LSD/main_on_linux.cpp:108-124)."""
    cells = np.arange(256, dtype=np.uint8).view(np.int8).reshape(16, 16)
    got = oracle.occupancy_to_map(cells)
    want = np.ones(256, np.uint8); want[0] = 255; want[255] = 0           # 0 free -> 255, -1 unknown -> 0, 1..100 (and the rest) -> 1
    assert np.array_equal(got.ravel(), want)
    assert got[cells == -1].tolist() == [0] and got[cells == 100].tolist() == [1]


def test_scan_to_map_match_restatement_finds_the_true_pose(maps, maps_meta, oracle, lsdmod):
    """Sanity of the (unpinned) restatement of myfa::thread_ScanToMapMatch: on a synthetic frame cut out of a fixture map the
    best-scoring candidate is the true pose."""
    from matching_case import build_case
    case = build_case(maps["aisle1"], maps_meta["aisle1"]["res"], oracle)
    pairs = lsdmod.match_pairs(case["map_lines"], case["scan_lines"])
    assert len(pairs) > 20
    sc = oracle.scan_to_map_match(case["map_cache"], case["map_lines"], case["scan_lines"], case["pts"], case["lidar"],
                                  (-1.0, -1.0, 0.0), pairs).reshape(-1, 4)
    fin = sc[np.isfinite(sc[:, 3])]
    best = fin[np.argmin(fin[:, 3])]
    assert best[3] < 0.2                                                    # mean distance (m) of the scan cells to occupied map cells
    assert abs(best[2] - case["theta"]) < 2.0 and np.hypot(*(best[:2] - case["lidar_map"])) < 3.0
    # a last pose far away rejects every candidate (rotateScanIm :330)
    far = oracle.scan_to_map_match(case["map_cache"], case["map_lines"], case["scan_lines"], case["pts"], case["lidar"],
                                   (5000.0, 5000.0, 0.0), pairs)
    assert np.isinf(far[..., 3]).all()


LIBM_TIES = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "libm_ties.json")))


@pytest.mark.parametrize("name", sorted(LIBM_TIES))
def test_libm_tie_images_document_the_one_caveat(name, oracle):
    """tests/golden/libm_ties.npz (written by make_libm_ties.py): ALL 35 images of the random campaigns (160 000 + 3 900 large ones,
    tools/campaign.py) on which the HIP path and the glibc-built restatement disagree, and the five with an NFA comparison below the
    campaign's margin floor (`near*`).  The restatement rebuilt on correctly rounded
    functions (oracle/cr_shim.cpp) differs from the glibc one in exactly the recorded way: one accept / reject decision on a
    structural tie that a 1-ulp libm difference turns; the same seeds, the same regions up to there."""
    img = np.load(os.path.join(os.path.dirname(__file__), "golden", "libm_ties.npz"))[name]
    t = LIBM_TIES[name]
    kw = t["params"]
    a = oracle.lsd(img.copy(), debug=True, **kw)
    b = oracle.lsd(img.copy(), debug=True, _lib=oracle.lib_cr(), **kw)
    assert len(b["lines"]) == t["lines_cr"]                           # what the HIP path gives too (test_parity_gpu.py)
    assert len(a["lines"]) == t["lines_glibc"], "this libm rounds differently from the glibc the campaign ran against"
    assert int((a["dbg"]["used"] != b["dbg"]["used"]).sum()) == t["used_diff"]
    assert int((a["lineIm"] != b["lineIm"]).sum()) == t["lineim_diff"]
    if name.startswith("near"):                                       # (an NFA comparison inside the libms' noise, or within a factor two of it -- below the
        #                                                                # floor tools/campaign.py enforces: the two builds decide alike on it)
        assert t["used_diff"] == 0 and t["lineim_diff"] == 0 and a["dbg"]["nfa_min_gap"] < 2.0
        return
    assert t["used_diff"] + t["lineim_diff"] > 0
    sa, sb = a["dbg"]["seeds"], b["dbg"]["seeds"]
    key = lambda s: (int(s["order_idx"]), int(s["num"]), int(s["final_num"]), int(s["outcome"]))
    first = next((i for i, (x, y) in enumerate(zip(sa, sb)) if key(x) != key(y)), None)
    if first is None:                                                 # every seed decided alike: one accepted rectangle was improved differently
        ra, rb = a["dbg"]["recs"], b["dbg"]["recs"]
        assert ra.shape == rb.shape and 1 <= int((np.abs(ra - rb).max(axis=1) > 1e-9).sum()) <= 2
    else:
        assert sa[first]["order_idx"] == sb[first]["order_idx"] and sa[first]["num"] == sb[first]["num"]   # the same seed grows the same region ...
        assert key(sa[first])[2:] != key(sb[first])[2:]                                                     # ... and is rated differently


def test_scan_to_map_match_without_scan_points_is_rejected(maps, oracle):
    """myFA.cpp:248-263: a rotated scan image without points (RSI.numScanImPoint == 0) is not scored, the candidate gets
    INFINITY -- not the 0/0 CalcScore's formula would give."""
    mc = np.ones((40, 50), np.float64)
    ln = np.zeros(1, oracle.LINE_DTYPE)
    ln["x1"], ln["y1"], ln["x2"], ln["y2"] = 5, 5, 30, 5
    out = oracle.scan_to_map_match(mc, ln, ln, np.zeros((0, 3)), (10.0, 10.0, 0.0), (-1.0, -1.0, 0.0), np.array([[0, 0]], np.int32))
    assert out.shape == (1, 4, 4) and np.all(np.isinf(out[0, :, 3])) and np.all(np.isfinite(out[0, :, :3]))


# ---- myrdp::FeatureScan --------------------------------------------------------------------------
RDP_MAP_PARAM = (1377, 428, 0.025, -4.43187, -5.49357)          # data/mapParam.txt (the map the lidar log belongs to)


def rdp_golden_check(res, z):
    """(lines of data/ScanlinesInfo.txt reproduced exactly, its raster pixels hit) for a FeatureScan result of the golden's frame."""
    gold = z["matlab_ScanlinesInfo"]
    L = res["lines"] if "lines" in res else res["linesInfo"]
    ours = np.stack([L[k] for k in ("k", "b", "dx", "dy", "x1", "y1", "x2", "y2", "len")], 1)
    matched = 0
    for row in gold:
        for o in ours:
            if np.array_equal(o[4:8], row[4:8]) and np.allclose(o, row, rtol=1e-13, atol=1e-13):
                matched += 1
                break
    im = res["lineIm"] > 0
    hits = 0
    for y, x in z["matlab_ScanlineIm_lit_yx"]:                   # the prototype's raster sits at (-1, -1) of the C++ coordinates
        if y + 1 < im.shape[0] and x + 1 < im.shape[1] and im[y + 1, x + 1]:
            hits += 1
    return matched, hits


def test_feature_scan_against_the_matlab_golden(oracle):
    """The only outputs the reference holds for FeatureScan: data/ScanlinesInfo.txt (12 lines) and data/ScanlineIm.txt (1003 lit
    pixels of a 237 x 832 raster) of the MATLAB prototype, for frame 31 of data/Lidar.txt (found by trying every frame: the only
    one that gives a 832 x 237 image).  The restatement yields 12 lines too; 8 of the golden 12 are reproduced with identical
    integer end points and k, b, dx, dy, len to 13 digits; the other four are the bottom wall, which the prototype splits at
    three more points (and it does not have two short segments the C++ code finds).  A loose pin, like the LSD one."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "lidar.npz"))
    sc = z["lidar"][int(z["matlab_frame"])]
    sc = sc[np.isfinite(sc[:, 0])]
    assert len(sc) == 326
    res = oracle.feature_scan(sc, RDP_MAP_PARAM)
    assert len(res["lines"]) == 12 and res["im_size"] == (832, 237) and tuple(z["matlab_ScanlineIm_shape"]) == (237, 832)
    matched, hits = rdp_golden_check(res, z)
    assert matched == 8
    assert hits >= 700                                            # measured: 711 of 1003


def test_feature_scan_on_the_whole_lidar_log(oracle):
    """Every frame of data/Lidar.txt: the raster pixels lie inside the image and off row / column 0 (the reference's "invalid" mark),
    the line count stays below the reference's 360 records, end points are integers inside the image."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "lidar.npz"))
    total = 0
    for f in range(len(z["lidar"])):
        sc = z["lidar"][f]
        sc = sc[np.isfinite(sc[:, 0])]
        res = oracle.feature_scan(sc, RDP_MAP_PARAM)
        w, h = res["im_size"]
        p = res["pts"]
        assert len(res["lines"]) < 360 and len(p) > 0
        assert p[:, 0].min() >= 1 and p[:, 0].max() < w and p[:, 1].min() >= 1 and p[:, 1].max() < h
        for k in ("x1", "y1", "x2", "y2"):
            assert np.array_equal(res["lines"][k], np.round(res["lines"][k]))
        total += len(res["lines"])
    assert total > 900
