"""Parity tests proper (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the
same inputs, against the committed golden answers, and -- at BASELINE.json's full size -- through
size-independent properties.

Stated tolerances (north_star: "line-for-line within a stated endpoint/angle tolerance, bit-exact
UsedMap indexing"):
  * GaussImage, magMap, maxGrad, sorted seed order, usedMap, lineIm, in-place remap, line count,
    orient: BIT-EXACT;
  * degMap: <= 1 ulp (the device atan2 is correctly rounded, glibc's is within 1 ulp of that);
  * logNFA of every seed that reaches RectangleImprover: BIT-EXACT against the restatement built on correctly rounded exp / log10 /
    pow (the device evaluates them correctly rounded: crmath.h), <= 4 ulp of max(|logNFA|, logNT) against the glibc build (glibc's
    log10 is off by one ulp in one call out of seven);
  * line endpoints x1,y1,x2,y2 and len: 1e-6 px absolute; dx,dy: 1e-9; k,b: 1e-6 relative
    (transcendentals on the rectangle path differ by ulps between OCML and glibc).
"""
import importlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import tile2048

pytestmark = pytest.mark.gpu

FIXTURES = ["map1", "mapValue", "aisle1", "aisle2", "aisle3", "f3key", "f4key"]
ENDPOINT_TOL = 1e-6
DIR_TOL = 1e-9
REL_TOL = 1e-6
DEG_ULP = 1
NFA_ULP = 4


@pytest.fixture(scope="module")
def ctx(lsdmod):
    c = lsdmod.Context(0)
    yield c
    c.close()


def ulps(a, b):
    return np.abs(a.view(np.int64) - b.view(np.int64))


def assert_lines_close(got, ref):
    assert len(got) == len(ref)
    if not len(ref):
        return
    for f in ("x1", "y1", "x2", "y2", "len"):
        assert np.abs(got[f] - ref[f]).max() <= ENDPOINT_TOL, f
    for f in ("dx", "dy"):
        assert np.abs(got[f] - ref[f]).max() <= DIR_TOL, f
    for f in ("k", "b"):
        a, b = got[f], ref[f]
        fin = np.isfinite(b)
        assert np.array_equal(np.isfinite(a), fin)
        assert np.array_equal(a[~fin], b[~fin], equal_nan=True)
        assert np.all(np.abs(a[fin] - b[fin]) <= REL_TOL * np.maximum(1.0, np.abs(b[fin]))), f
    assert np.array_equal(got["orient"], ref["orient"])


def full_check(lsdmod, ctx, oracle, img, params=None, kw=None):
    kw = kw or {}
    ref_map = img.copy()
    ref = oracle.lsd(ref_map, debug=True, **kw)
    d = ref["dbg"]
    w, h = d["w"], d["h"]
    got_map = img.copy()
    lines, line_im = ctx.run(got_map, params)
    assert np.array_equal(got_map, ref_map)                                   # observable in-place remap (Q2)
    assert np.array_equal(ctx.fetch(0, lsdmod.DBG_GAUSS, (w, h)), d["gauss"])
    assert np.array_equal(ctx.fetch(0, lsdmod.DBG_MAG, (w, h)), d["mag"])
    assert ctx.fetch(0, lsdmod.DBG_MAXGRAD, (w, h)) == d["maxGrad"]
    assert ulps(ctx.fetch(0, lsdmod.DBG_DEG, (w, h)), d["deg"]).max() <= DEG_ULP
    order = ctx.fetch(0, lsdmod.DBG_ORDER, (w, h)).astype(np.int64)
    assert len(order) == d["nb"]
    assert np.array_equal(order, d["ord_y"].astype(np.int64) * w + d["ord_x"])
    assert np.array_equal(ctx.fetch(0, lsdmod.DBG_ORDER_VAL, (w, h)), d["ord_v"])
    used = (ctx.fetch(0, lsdmod.DBG_STATE, (w, h)) & 3).astype(np.uint8)
    assert np.array_equal(used, d["used"])                                     # bit-exact UsedMap
    assert np.array_equal(line_im, ref["lineIm"])
    assert_lines_close(lines, ref["lines"])
    st = ctx.fetch(0, lsdmod.DBG_STATS, (w, h))
    # work counters include speculative evaluations that were discarded, so they bound the oracle's from above
    for k in ("grow_calls", "grown_px", "nfa_calls"):
        assert st[k] >= d[k], k
    assert st["rrr_oob_reads"] == 0 == d["rrr_oob_reads"]
    return lines, line_im, ref


@pytest.mark.parametrize("name", FIXTURES)
def test_fixture_parity(name, maps, lsdmod, ctx, oracle):
    full_check(lsdmod, ctx, oracle, maps[name])


@pytest.mark.parametrize("name", FIXTURES)
def test_fixture_known_answers(name, maps, known, lsdmod, ctx):
    """The HIP path against the reference's recorded outputs directly (no oracle in the loop)."""
    nl, lit = known["counts"][name][:2]
    lines, line_im = ctx.run(maps[name].copy())
    assert len(lines) == nl
    assert int((line_im == 255).sum()) == lit


def test_map1_golden_line_list(maps, known, lsdmod, ctx):
    lines, _ = ctx.run(maps["map1"].copy())
    gold = known["map1_lines"]
    assert len(lines) == len(gold)
    for got, row in zip(lines, gold):
        for f, s in zip(known["map1_lines_fields"], row):
            if f == "orient":
                assert int(got[f]) == int(s)
            elif f in ("k", "b"):
                assert abs(got[f] - float(s)) <= REL_TOL * max(1.0, abs(float(s)))
            elif f in ("dx", "dy"):
                assert abs(got[f] - float(s)) <= DIR_TOL
            else:
                assert abs(got[f] - float(s)) <= ENDPOINT_TOL


def test_matlab_golden_files_directly(maps, lsdmod, ctx):
    """The HIP path against the only outputs the reference itself holds for this path (data/MaplinesInfo.txt and
    data/MaplineIm.txt for data/mapValue.txt) -- no oracle in the loop; same bounds as the oracle's own pin
    (tests/test_oracle.py::test_matlab_golden_pins_the_oracle): 36 of 40 lines within 0.1 px, 3865 of 3992 raster pixels."""
    from test_oracle import matlab_golden_check
    lines, line_im = ctx.run(maps["mapValue"].copy())
    matched, hits = matlab_golden_check(lines, line_im, maps)
    assert (matched, hits) == (36, 3865)


def test_front_end_only_on_map1(maps, lsdmod, ctx, oracle):
    """BASELINE.json configs[1]: mapValue_map1 with the FRONT END on the GPU only (remap + Gaussian + gradient / level-line angle /
    pseudo-ordering; lsd_set_stop_after) -- what a caller that keeps the region stage on the host would take from the library.
    Every array the region stage reads is the oracle's: GaussImage, magMap, maxGrad and the sorted seed list bit for bit, degMap
    to 1 ulp, the threshold flags; no region code has run (no pixel of code 2 or 3, no line)."""
    img = maps["map1"]
    ref_map = img.copy()
    d = oracle.lsd(ref_map, debug=True)["dbg"]
    w, h = d["w"], d["h"]
    try:
        ctx.set_stop_after(lsdmod.STAGE_SORT)
        got_map = img.copy()
        lines, line_im = ctx.run(got_map)
        assert len(lines) == 0 and not line_im.any()
        assert np.array_equal(got_map, ref_map)                               # the in-place remap is part of the front end (myLSD.cpp:135-142)
        assert np.array_equal(ctx.fetch(0, lsdmod.DBG_GAUSS, (w, h)), d["gauss"])
        assert np.array_equal(ctx.fetch(0, lsdmod.DBG_MAG, (w, h)), d["mag"])
        assert ctx.fetch(0, lsdmod.DBG_MAXGRAD, (w, h)) == d["maxGrad"]
        assert ulps(ctx.fetch(0, lsdmod.DBG_DEG, (w, h)), d["deg"]).max() <= DEG_ULP
        order = ctx.fetch(0, lsdmod.DBG_ORDER, (w, h)).astype(np.int64)
        assert len(order) == d["nb"] == ctx.fetch(0, lsdmod.DBG_NB, (w, h))
        assert np.array_equal(order, d["ord_y"].astype(np.int64) * w + d["ord_x"])
        assert np.array_equal(ctx.fetch(0, lsdmod.DBG_ORDER_VAL, (w, h)), d["ord_v"])
        code = (ctx.fetch(0, lsdmod.DBG_STATE, (w, h)) & 3).astype(np.uint8)
        thr = 2.0 / np.sin(22.5 / 180.0 * np.pi)                                # myLSD.cpp:148-149, :165
        below = d["mag"] < thr
        below[0, :] = False; below[:, 0] = False                                # row 0 / column 0 stay 0 (Q3)
        assert np.array_equal(code == 1, below) and not (code >= 2).any()
        # ... and the gradient stage alone (stop after 2): the same Gaussian and magnitudes, no seed list asked for
        ctx.set_stop_after(lsdmod.STAGE_GRAD)
        ctx.run(img.copy())
        assert np.array_equal(ctx.fetch(0, lsdmod.DBG_MAG, (w, h)), d["mag"])
        assert ctx.fetch(0, lsdmod.DBG_MAXGRAD, (w, h)) == d["maxGrad"]
    finally:
        ctx.set_stop_after(lsdmod.STAGE_ALL)
    lines, _ = ctx.run(img.copy())                                            # the switch leaves nothing behind
    assert len(lines) == 7


def test_reference_maps_as_batches_match_oracle(maps, lsdmod, ctx, oracle):
    """bench.py's `real_maps` workload (the reference's seven maps as they are, LSD/main_on_windows.cpp:20-46, and the 2048^2 canvas of
    three different maps pasted side by side), each as a resident batch of copies through the device entry point: every copy's line
    count, records and lineIm are the oracle's for that map."""
    import torch
    import bench
    reps = 96
    todo = [(k, maps[k]) for k in bench.REAL_MAPS] + [("pasted2048", bench.make_pasted(maps, 2048))]
    for name, im in todo:
        ref = oracle.lsd(im.copy())
        rows, cols = im.shape
        d_maps = torch.from_numpy(np.broadcast_to(im, (reps, rows, cols)).copy()).cuda()
        max_lines = 512
        d_lines = torch.zeros((reps, max_lines, 10), dtype=torch.int64, device="cuda")
        d_counts = torch.zeros(reps, dtype=torch.int32, device="cuda")
        d_ims = torch.zeros((reps, rows, cols), dtype=torch.uint8, device="cuda")
        ctx.enqueue_device(d_maps.data_ptr(), reps, cols, rows, d_lines.data_ptr(), max_lines, d_counts.data_ptr(),
                           d_line_ims=d_ims.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        counts = d_counts.cpu().numpy()
        assert (counts == len(ref["lines"])).all(), (name, counts.min(), counts.max(), len(ref["lines"]))
        ref_im = torch.from_numpy(ref["lineIm"]).cuda()
        assert bool((d_ims == ref_im[None]).all().item()), name
        raw = d_lines[:, :counts[0]].cpu().numpy()
        got0 = raw[0].copy().view(np.uint8).reshape(-1, 80).view(lsdmod.LINE_DTYPE).reshape(-1)
        assert_lines_close(got0, ref["lines"])
        assert all(raw[i].tobytes() == raw[0].tobytes() for i in range(1, reps)), name
        del d_maps, d_lines, d_counts, d_ims


def test_tile2048_parity_and_known(maps, known, lsdmod, ctx, oracle):
    img = tile2048(maps["aisle1"])
    lines, line_im, _ = full_check(lsdmod, ctx, oracle, img)
    assert len(lines) == known["counts"]["tile2048"][0]
    assert int((line_im == 255).sum()) == known["counts"]["tile2048"][1]


def test_seed_trace_matches_oracle(maps, lsdmod, ctx, oracle):
    """Every seed makes the same decision (small / refine-failed / NFA-rejected / accepted) with the same region sizes."""
    img = maps["f4key"]
    ref = oracle.lsd(img.copy(), debug=True)["dbg"]
    ctx.set_trace(True)
    try:
        ctx.run(img.copy())
        seeds = ctx.fetch(0, lsdmod.DBG_SEEDS, (ref["w"], ref["h"]))
    finally:
        ctx.set_trace(False)
    rs = ref["seeds"]
    assert len(seeds) == len(rs)
    for f in ("order_idx", "x", "y", "num", "outcome", "final_num"):
        assert np.array_equal(seeds[f], rs[f]), f
    ev = rs["outcome"] >= 2                                                   # reached RectangleImprover (:240)
    assert ev.sum() > 100
    a, b = np.ascontiguousarray(seeds["logNFA"][ev]), np.ascontiguousarray(rs["logNFA"][ev])
    fin = np.isfinite(b)
    assert np.array_equal(a[~fin], b[~fin])
    logNT = 5 * (np.log10(ref["h"]) + np.log10(ref["w"])) / 2.0
    assert np.all(np.abs(a[fin] - b[fin]) <= NFA_ULP * np.spacing(np.maximum(np.abs(b[fin]), logNT)))
    assert not seeds["logNFA"][~ev].any() and not rs["logNFA"][~ev].any()
    # ... and to the bit against the same restatement on correctly rounded exp / log10 / pow (and sin / cos / atan2): oracle/cr_shim.cpp
    rc = oracle.lsd(img.copy(), debug=True, _lib=oracle.lib_cr())["dbg"]["seeds"]
    assert len(rc) == len(seeds) and np.array_equal(seeds["outcome"], rc["outcome"])
    assert np.array_equal(seeds["logNFA"], rc["logNFA"])


def test_fast_sincos_error_bound(lsdmod, ctx):
    """RegionGrower's classifier works on fp32 ESTIMATES (k_region.hip): the unit vector of a pixel's packed angle -- fp32 with
    the two lowest mantissa bits dropped, hardware sin/cos -- must lie within 3e-6 of the exact one (kEpsU = 6e-6 in k_region.hip = this + the 1.9e-6 of a batch's fp32 partial sums, bounded below); every margin
    of the classifier is built on that bound (a candidate closer to the tolerance than the margins goes to the exact test)."""
    rng = np.random.default_rng(11)
    a = np.concatenate([rng.uniform(-np.pi, np.pi, 4_000_000), np.linspace(-np.pi, np.pi, 200_001),
                        np.array([0.0, np.pi, -np.pi, np.pi / 2, -np.pi / 2, 1e-30, -1e-30, np.nextafter(np.pi, 0)]),
                        rng.uniform(-1e-3, 1e-3, 100_000)])
    s, c = ctx.eval_math(3, a)
    err = np.hypot(s - np.sin(a), c - np.cos(a))
    assert err.max() <= 3.0e-6, err.max()                                      # the first part of kEpsU = 6e-6 (k_region.hip)
    # the second part: 64 such unit vectors summed in fp32 (a batch's partial sums) against the fp64 sum of the same terms
    rng2 = np.random.default_rng(7)
    worst = 0.0
    for _ in range(200):
        base = rng2.uniform(-np.pi, np.pi)
        a = base + rng2.uniform(-0.4, 0.4, 64)                                 # (one region: directions within the tolerance)
        s_, c_ = ctx.eval_math(3, a)
        for v in (s_, c_):
            acc = np.float32(0.0)
            for t in v.astype(np.float32):
                acc = np.float32(acc + t)
            worst = max(worst, abs(float(acc) - float(v.astype(np.float64).sum())) / 64.0)
    assert worst <= 1.9e-6, worst                                              # per term: <= 2^-24 * 32


def test_reference_names(maps, lsdmod, ctx, oracle):
    """myLineSegmentDetector / runLSD / structLSD as the reference spells them (LSD/myLSD.h:123-132)."""
    img = maps["map1"]
    m = img.copy()
    r = lsdmod.myLineSegmentDetector(m, img.shape[1], img.shape[0], lsdmod.lsd_sca, lsdmod.lsd_sig, lsdmod.lsd_angThre,
                                     lsdmod.lsd_denThre, lsdmod.pseBin, ctx=ctx)
    assert r.len_linesInfo == 7 and r.lineIm.shape == img.shape and r.lineIm.dtype == np.uint8
    r2 = lsdmod.runLSD(img.copy(), ctx=ctx)
    assert r2.linesInfo.tobytes() == r.linesInfo.tobytes()
    ref = oracle.lsd(img.copy())
    assert_lines_close(r.linesInfo, ref["lines"])


def test_batch_equals_single(maps, lsdmod, ctx):
    src = maps["aisle2"][:600, :1600]
    batch = np.stack([src, src[::-1].copy(), src[:, ::-1].copy(), np.roll(src, 37, 1)]).copy()
    lines, offs, ims = ctx.run_batch(batch.copy())
    assert offs[0] == 0 and len(offs) == 5
    for i in range(4):
        l1, im1 = ctx.run(batch[i].copy())
        assert lines[offs[i]:offs[i + 1]].tobytes() == l1.tobytes()
        assert np.array_equal(ims[i], im1)


def test_counter_records_in_one_copy(maps, lsdmod, ctx):
    """lsd_debug_fetch(LSD_DBG_STATS) with room for several images returns the records of the following images as well
    (what the developer probes read a whole batch with): the same words as one fetch per image."""
    src = maps["aisle2"][:600, :1600]
    batch = np.stack([src, src[::-1].copy(), src[:, ::-1].copy()]).copy()
    ctx.run_batch(batch.copy())
    wh = lsdmod.scaled_size(1600, 600)
    block = ctx.fetch_stats_block(3)
    assert block.shape == (3, 48)
    for i in range(3):
        one = ctx.fetch(i, lsdmod.DBG_STATS, wh)
        assert one["grow_calls"] == block[i, 0] and one["grown_px"] == block[i, 1] and one["nfa_calls"] == block[i, 2]
        assert one["cycles_total"] == block[i, 8] and one["seeds"] == block[i, 15] and block[i, 15] > 0


def test_deterministic(maps, lsdmod, ctx):
    a = ctx.run(maps["aisle3"].copy())
    b = ctx.run(maps["aisle3"].copy())
    assert a[0].tobytes() == b[0].tobytes() and np.array_equal(a[1], b[1])


def test_batch_is_deterministic_run_to_run(maps, lsdmod, ctx):
    """Speculation, the commit order and waves finishing in different orders must not show: the same 48-image batch (one
    image per compute unit, all at once) gives byte-identical records five times in a row."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    batch = bench.make_batch(maps, 48, 1024)
    ref = None
    try:
        for rep in range(12):                                      # both builds of the region stage in turn (certified sets are founded by
            ctx.set_region_waves(4 if rep % 2 else 0)              # whichever evaluation gets there first: the results must not show it)
            lines, offs, ims = ctx.run_batch(batch.copy())
            cur = (lines.tobytes(), offs.tobytes(), ims.tobytes())
            if ref is None:
                ref = cur
            assert cur == ref, rep
    finally:
        ctx.set_region_waves(0)


def test_non_packed_stride(maps, lsdmod, ctx, oracle):
    img = maps["map1"]
    rows, cols = img.shape
    big = np.zeros((rows, cols + 24), np.uint8)
    big[:, :cols] = img
    view = big[:, :cols]
    p = lsdmod.make_params()
    import ctypes as C
    line_im = np.zeros((rows, cols + 8), np.uint8)
    lines_p, n = C.c_void_p(), C.c_int()
    st = ctx.L.lsd_run(ctx.h, view.ctypes.data, cols, rows, big.strides[0], C.byref(p), line_im.ctypes.data,
                       line_im.strides[0], C.byref(lines_p), C.byref(n))
    assert st == 0 and n.value == 7
    ctx.L.lsd_free(lines_p)
    ref_map = img.copy()
    ref = oracle.lsd(ref_map)
    assert np.array_equal(big[:, :cols], ref_map) and not big[:, cols:].any()
    assert np.array_equal(line_im[:, :cols], ref["lineIm"]) and not line_im[:, cols:].any()


@pytest.mark.parametrize("kw", [dict(sig=0.8), dict(angThre=20.0), dict(denThre=0.6), dict(pseBin=512),
                                dict(sig=0.45, angThre=30.0, denThre=0.8, pseBin=256)])
def test_other_parameters(kw, maps, lsdmod, ctx, oracle):
    full = dict(sca=0.3, sig=0.6, angThre=22.5, denThre=0.7, pseBin=1024)
    full.update(kw)
    full_check(lsdmod, ctx, oracle, maps["mapValue"], lsdmod.make_params(**full), kw=full)


def synth(seed, rows, cols):
    rng = np.random.default_rng(seed)
    m = np.zeros((rows, cols), np.uint8)
    m[rng.random((rows, cols)) < 0.35] = 255                                  # unknown cells
    for _ in range(12):                                                       # walls: axis-aligned and slanted
        x0, y0 = rng.integers(5, cols - 5), rng.integers(5, rows - 5)
        L = int(rng.integers(30, 200)); a = rng.choice([0, np.pi / 2, rng.uniform(0, np.pi)])
        t = np.arange(L)
        xs = np.clip((x0 + t * np.cos(a)).astype(int), 0, cols - 1); ys = np.clip((y0 + t * np.sin(a)).astype(int), 0, rows - 1)
        m[ys, xs] = 1
    return m


@pytest.mark.parametrize("seed,rows,cols", [(1, 240, 320), (2, 333, 517), (3, 101, 999), (4, 700, 64)])
def test_synthetic_ragged_sizes(seed, rows, cols, lsdmod, ctx, oracle):
    """Odd sizes (tiles that straddle the border, w or h not multiples of anything), exact ties from
    axis-aligned walls (NaN/inf rectangle slopes)."""
    full_check(lsdmod, ctx, oracle, synth(seed, rows, cols))


def test_blank_and_minimum_images(lsdmod, ctx, oracle):
    lines, im = ctx.run(np.zeros((64, 80), np.uint8))
    assert len(lines) == 0 and not im.any()
    # all-unknown map: row 0 / col 0 keep 255 while the interior is remapped to 0 (Q2) -> a border edge
    full_check(lsdmod, ctx, oracle, np.full((40, 40), 255, np.uint8))
    with pytest.raises(lsdmod.LsdError) as e:                                 # scaled size < 2: nothing to do
        ctx.run(np.zeros((5, 5), np.uint8))
    assert e.value.status == lsdmod.LSD_ERR_INVALID
    with pytest.raises(lsdmod.LsdError) as e:
        ctx.run(np.zeros((64, 64), np.uint8), lsdmod.make_params(pseBin=4096))
    assert e.value.status == lsdmod.LSD_ERR_UNSUPPORTED


def test_border_pixels_keep_raw_values(maps, lsdmod, ctx, oracle):
    img = maps["map1"].copy()
    img[0, 10:200] = 1; img[5:300, 0] = 255; img[0, 0] = 1
    full_check(lsdmod, ctx, oracle, img)


def test_line_capacity_is_reported(maps, lsdmod, ctx):
    import torch
    img = torch.from_numpy(maps["aisle1"].copy()).cuda()
    rows, cols = img.shape
    lines = torch.zeros((1, 8, 10), dtype=torch.int64, device="cuda")
    counts = torch.zeros(1, dtype=torch.int32, device="cuda")
    ctx.enqueue_device(img.data_ptr(), 1, cols, rows, lines.data_ptr(), 8, counts.data_ptr(),
                       stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int(counts[0]) == 75                                              # true count, only 8 stored


def test_device_resident_batch_with_torch(maps, lsdmod, ctx, oracle):
    """The device entry point on torch-owned HBM buffers and torch's stream (the bench's path)."""
    import torch
    src = maps["aisle3"]
    batch = np.stack([src, np.roll(src, 11, 0), src[::-1].copy()]).copy()
    n, rows, cols = batch.shape
    d_maps = torch.from_numpy(batch).cuda()
    max_lines = 512
    d_lines = torch.zeros((n, max_lines, 10), dtype=torch.int64, device="cuda")
    d_counts = torch.zeros(n, dtype=torch.int32, device="cuda")
    d_ims = torch.zeros((n, rows, cols), dtype=torch.uint8, device="cuda")
    ctx.enqueue_device(d_maps.data_ptr(), n, cols, rows, d_lines.data_ptr(), max_lines, d_counts.data_ptr(),
                       d_line_ims=d_ims.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(d_maps.cpu().numpy(), batch)                        # read-only without the write-back flag
    counts = d_counts.cpu().numpy()
    raw = d_lines.cpu().numpy()
    for i in range(n):
        ref = oracle.lsd(batch[i].copy())
        assert counts[i] == len(ref["lines"])
        got = raw[i, :counts[i]].copy().view(np.uint8).reshape(-1, 80).view(lsdmod.LINE_DTYPE).reshape(-1)
        assert_lines_close(got, ref["lines"])
        assert np.array_equal(d_ims[i].cpu().numpy(), ref["lineIm"])


def test_line_raster_at_an_unaligned_address(maps, lsdmod, ctx, oracle):
    """lineIm is cleared by the library's own 16-byte-per-lane kernel (k_clear16): a raster that starts at an odd address and has an odd
    size gets its head and tail bytes too, and nothing outside it is touched."""
    import torch
    img = np.ascontiguousarray(maps["aisle2"][:301, :455])
    rows, cols = img.shape
    n = 3
    d = torch.from_numpy(np.stack([img] * n)).cuda()
    pad = 37
    buf = torch.full((n * rows * cols + 2 * pad + 16,), 0xAB, dtype=torch.uint8, device="cuda")
    base = buf.data_ptr() + pad + ((3 - (buf.data_ptr() + pad)) % 16)          # address = 3 (mod 16)
    off = base - buf.data_ptr()
    lines = torch.zeros((n, 256, 10), dtype=torch.int64, device="cuda"); cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
    ctx.enqueue_device(d.data_ptr(), n, cols, rows, lines.data_ptr(), 256, cnt.data_ptr(), d_line_ims=base, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ref = oracle.lsd(img.copy())
    got = buf.cpu().numpy()
    assert (got[:off] == 0xAB).all() and (got[off + n * rows * cols:] == 0xAB).all()
    for i in range(n):
        assert np.array_equal(got[off + i * rows * cols: off + (i + 1) * rows * cols].reshape(rows, cols), ref["lineIm"]), i


def test_full_size_batch_properties(maps, lsdmod, ctx, oracle):
    """BASELINE config 4 shape (2048x2048 tiles), a slice of the batch: image 0 is the unshifted aisle1
    tile whose reference answer is recorded; replicas must agree bit for bit; every output is a valid raster."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    batch = bench.make_batch(maps, 6, 2048, first=0)
    batch = np.concatenate([batch, batch[:2]])                                # replicas of images 0 and 1
    lines, offs, ims = ctx.run_batch(batch.copy())
    cnt = np.diff(offs)
    assert cnt[0] == 238 and int((ims[0] == 255).sum()) == 20296              # SURVEY 8c tile2048
    assert cnt[6] == cnt[0] and cnt[7] == cnt[1]
    assert lines[offs[6]:offs[7]].tobytes() == lines[offs[0]:offs[1]].tobytes()
    assert lines[offs[7]:offs[8]].tobytes() == lines[offs[1]:offs[2]].tobytes()
    assert np.array_equal(ims[6], ims[0]) and np.array_equal(ims[7], ims[1])
    assert set(np.unique(ims)) <= {0, 255}
    for i in (1, 3):                                                          # spot-check two against the oracle
        ref = oracle.lsd(batch[i].copy())
        assert_lines_close(lines[offs[i]:offs[i + 1]], ref["lines"])
        assert np.array_equal(ims[i], ref["lineIm"])
    for i in range(8):                                                        # every line lies inside the canvas, len consistent
        L = lines[offs[i]:offs[i + 1]]
        assert np.all(L["len"] > 0) and np.all(np.abs(np.hypot(L["x2"] - L["x1"], L["y2"] - L["y1"]) - L["len"]) < 1e-9)


def test_bench_batch_sample_matches_oracle(maps, lsdmod, ctx, oracle):
    """A 32-image slice of the bench batch (2048x2048, all four sources, all flips, the heaviest 'mapValue' tiles included)
    run as ONE batch -- every wavefront of the region stage speculating at once -- against the oracle image by image."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    ids = list(range(16)) + [27, 151, 155, 183, 187, 203, 244, 271, 347, 355, 383, 391, 419, 496, 51, 55]
    batch = np.stack([bench.make_image(maps, i, 2048) for i in ids])
    lines, offs, ims = ctx.run_batch(batch.copy())
    for j, i in enumerate(ids):
        ref = oracle.lsd(batch[j].copy())
        assert offs[j + 1] - offs[j] == len(ref["lines"]), i
        assert_lines_close(lines[offs[j]:offs[j + 1]], ref["lines"])
        assert np.array_equal(ims[j], ref["lineIm"]), i


def test_whole_bench_batch_matches_oracle(maps, lsdmod, ctx, oracle):
    """BASELINE.json's full configuration -- the 512 x 2048x2048 batch bench.py times -- through the device entry point in ONE
    launch sequence (the 8-wavefront region stage the library picks for this batch size, every workgroup of the GPU speculating at once): every image's line count,
    lineIm and line records against the oracle.  (~20 s, most of it the oracle.)"""
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    n, size = 512, 2048
    host = bench.make_batch(maps, n, size)
    d = torch.from_numpy(host).cuda()
    d_lines = torch.zeros((n, 1024, 10), dtype=torch.int64, device="cuda")
    d_counts = torch.zeros(n, dtype=torch.int32, device="cuda")
    d_ims = torch.zeros((n, size, size), dtype=torch.uint8, device="cuda")
    ctx.enqueue_device(d.data_ptr(), n, size, size, d_lines.data_ptr(), 1024, d_counts.data_ptr(), d_line_ims=d_ims.data_ptr(),
                       stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    cnt = d_counts.cpu().numpy()
    rec = d_lines.cpu().numpy().view(np.uint8).reshape(n, 1024, 80)
    assert cnt.max() <= 1024
    total = 0
    for i in range(n):
        ref = oracle.lsd(host[i].copy())
        assert cnt[i] == len(ref["lines"]), i
        assert np.array_equal(d_ims[i].cpu().numpy(), ref["lineIm"]), i
        assert_lines_close(rec[i, :cnt[i]].copy().view(lsdmod.LINE_DTYPE).reshape(-1), ref["lines"])
        total += int(cnt[i])
    assert total == 138815                                                     # lines_per_step of the bench line
    del d, d_lines, d_ims
    torch.cuda.empty_cache()


def test_timed_configuration_of_the_bench_matches_oracle(maps, lsdmod, oracle):
    """The configuration bench.py's `value` is timed on, exactly: the 512 x 2048x2048 batch, EIGHT contexts / streams / output sets
    in flight (bench.py --pipeline 8, GPU_MAX_HW_QUEUES=16 from conftest, as in bench.py), the 4-wavefront region stage (lsd_set_region_waves 4: three
    workgroups per CU), help across workgroups off (lsd_set_region_help 0), no LSD_FLAG_WRITEBACK_MAP (the steps share one resident
    input), three rounds of eight steps.
    Context 0 of the last round is compared with the oracle image by image (counts, line records, lineIm); every other
    (round, context) result must equal it byte for byte -- counts, the valid line records and lineIm."""
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    n, size, depth, cap = 512, 2048, 8, 1024
    host = bench.make_batch(maps, n, size)
    d = torch.from_numpy(host).cuda()
    ctxs = [lsdmod.Context(0) for _ in range(depth)]
    streams = [torch.cuda.Stream() for _ in range(depth)]
    outs = [(torch.zeros((n, cap, 10), dtype=torch.int64, device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda"),
             torch.zeros((n, size, size), dtype=torch.uint8, device="cuda")) for _ in range(depth)]
    keep = None
    try:
        for c in ctxs:
            c.set_region_help(0)
            c.set_region_waves(4)
            c.reserve(n, size, size)
        for rnd in range(3):
            for o in outs:
                for t in o: t.fill_(7)                     # (stale results of the round before cannot pass for new ones)
            torch.cuda.synchronize()
            for c, o, st in zip(ctxs, outs, streams):      # the steps are enqueued back to back, as bench.py's timed loop does
                c.enqueue_device(d.data_ptr(), n, size, size, o[0].data_ptr(), cap, o[1].data_ptr(), d_line_ims=o[2].data_ptr(), stream=st.cuda_stream)
            torch.cuda.synchronize()
            assert torch.equal(d.cpu(), torch.from_numpy(host))                      # no write-back: the shared input is untouched
            cnt0 = outs[0][1]
            assert int(cnt0.min()) >= 0 and int(cnt0.max()) <= cap
            valid = (torch.arange(cap, device="cuda")[None, :] < cnt0[:, None])[:, :, None]   # records below an image's count
            for j in range(1, depth):
                assert torch.equal(outs[j][1], cnt0), (rnd, j)
                assert torch.equal(outs[j][0] * valid, outs[0][0] * valid), (rnd, j)
                assert torch.equal(outs[j][2], outs[0][2]), (rnd, j)
            if keep is None:
                keep = [t.clone() for t in outs[0]]
            else:
                assert torch.equal(keep[1], cnt0) and torch.equal(keep[0] * valid, outs[0][0] * valid) and torch.equal(keep[2], outs[0][2]), rnd
        cnt = outs[0][1].cpu().numpy()
        rec = outs[0][0].cpu().numpy().view(np.uint8).reshape(n, cap, 80)
        total = 0
        for i in range(n):
            ref = oracle.lsd(host[i].copy())
            assert cnt[i] == len(ref["lines"]), i
            assert np.array_equal(outs[0][2][i].cpu().numpy(), ref["lineIm"]), i
            assert_lines_close(rec[i, :cnt[i]].copy().view(lsdmod.LINE_DTYPE).reshape(-1), ref["lines"])
            total += int(cnt[i])
        assert total == 138815                                                 # lines_per_step of the bench line
    finally:
        for c in ctxs: c.close()
        del d, outs, keep
        torch.cuda.empty_cache()


def test_bench_strong_scaling_split(maps, lsdmod, ctx):
    """bench.py --scaling strong (BASELINE configs[4]: the SAME batch split over the ranks by dist.shard_range) with world size 1:
    the JSON line carries the whole batch's answer."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--scaling", "strong", "--batch", "6", "--size", "1024",
                        "--steps", "1", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["scaling"] == "strong" and j["config"]["images_total"] == 6 and j["n_gpus"] == 1
    lines, offs, _ = ctx.run_batch(bench.make_batch(maps, 6, 1024))
    assert j["lines_per_step"] == len(lines) == offs[-1]
    assert j["roofline"]["kernel"] == "k_gradient" and j["dominant_kernel"]["name"] == "k_region"
    assert j["dominant_kernel"]["cycles_per_image"]["max_over_mean"] >= 1.0 and len(j["dominant_kernel"]["cycles_per_image"]["bad_records"]) <= 1, j["dominant_kernel"]["cycles_per_image"]
    ldist = importlib.import_module("linesegmentdetector-slam_amd.dist")
    assert [ldist.shard_range(6, 4, r) for r in range(4)] == [(0, 2), (2, 3), (3, 5), (5, 6)]    # what each of 4 ranks would take


def test_cpp_adapter_runs(maps, tmp_path):
    """The C++ host side (include/myLSD.h) end to end: same line count as the recorded reference answer."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "linesegmentdetector-slam_amd")
    img = maps["map1"]
    raw = tmp_path / "map1.raw"
    img.tofile(raw)
    src = tmp_path / "t.cpp"
    src.write_text(
        '#include "myLSD.h"\n#include <cstdio>\n'
        'int main(int argc, char** argv) {\n'
        '  mylsd::Mat m = mylsd::make_u8(480, 608);\n'
        '  FILE* f = fopen(argv[1], "rb"); size_t n = fread(m.ptr<unsigned char>(0), 1, 480*608, f); fclose(f);\n'
        '  mylsd::MatF64 mc = mylsd::createMapCache(m, 0.05);   /* before the LSD call: it rewrites m (Q2) */\n'
        '  double msum = 0; long below = 0; for (int y = 0; y < 480; y++) for (int x = 0; x < 608; x++) { double v = mc.ptr<double>(y)[x]; msum += v; below += v < 1.0; }\n'
        '  mylsd::structLSD r = mylsd::myLineSegmentDetector(m, 608, 480, lsd_sca, lsd_sig, lsd_angThre, lsd_denThre, pseBin);\n'
        '  long lit = 0; for (int y = 0; y < 480; y++) for (int x = 0; x < 608; x++) lit += r.lineIm.ptr<unsigned char>(y)[x] == 255;\n'
        '  std::printf("%zu %d %ld %.17g %d %.6f %ld\\n", n, r.len_linesInfo, lit, r.linesInfo[0].x1, (int)m.ptr<unsigned char>(3)[3], msum, below);\n'
        '  int bad = 0; try { mylsd::myLineSegmentDetector(mylsd::make_u8(2, 2), 2, 2, 0.3, 0.6, 22.5, 0.7, 1024); } catch (const mylsd::lsd_error& e) { bad = e.status; }\n'
        '  std::printf("%d\\n", bad);\n'
        '  free(r.linesInfo); return 0; }\n')
    exe = tmp_path / "t"
    subprocess.run(["g++", "-std=c++17", "-I", os.path.join(root, "include"), str(src), "-o", str(exe),
                    "-L", pkg, "-llsdhip", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    p = subprocess.run([str(exe), str(raw)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    n, nl, lit, x1, _, msum, below, bad = p.stdout.split()
    assert int(n) == 480 * 608 and int(nl) == 7 and int(lit) == 545
    assert abs(float(x1) - 351.00301936775691) < ENDPOINT_TOL
    assert abs(float(msum) - 281563.372994) < 1e-5 and int(below) == 20064        # SURVEY 8c, map1 mapCache
    assert int(bad) == 1                                                           # LSD_ERR_INVALID surfaces as lsd_error


# ---- createMapCache: integer flood order + exactly rounded sqrt -> BIT-EXACT --------------
@pytest.mark.parametrize("name", FIXTURES)
def test_map_cache_fixture_parity(name, maps, maps_meta, known, lsdmod, ctx, oracle):
    img = maps[name]
    res = maps_meta[name]["res"]
    got = ctx.map_cache(img, res)
    ref = oracle.map_cache(img.copy(), res)
    assert np.array_equal(got, ref)
    csum, ccnt = known["counts"][name][2:]
    assert abs(float(got.sum()) - csum) < 1e-5 and int((got < 1.0).sum()) == ccnt   # the reference's recorded answers


def test_map_cache_tile2048_and_reference_name(maps, known, lsdmod, ctx, oracle):
    img = tile2048(maps["aisle1"])
    got = lsdmod.createMapCache(img, 0.025, ctx=ctx)
    assert np.array_equal(got, oracle.map_cache(img.copy(), 0.025))
    csum, ccnt = known["counts"]["tile2048"][2:]
    assert abs(float(got.sum()) - csum) < 1e-4 and int((got < 1.0).sum()) == ccnt


@pytest.mark.parametrize("seed,rows,cols,res", [(1, 97, 131, 0.05), (2, 300, 40, 0.025), (3, 64, 64, 0.2), (4, 5, 700, 0.1)])
def test_map_cache_synthetic(seed, rows, cols, res, lsdmod, ctx, oracle):
    rng = np.random.default_rng(seed)
    m = np.zeros((rows, cols), np.uint8)
    m[rng.random((rows, cols)) < 0.02] = 1
    m[rng.random((rows, cols)) < 0.3] = 255
    m[0, 0] = 1; m[rows - 1, cols - 1] = 1                                   # sources on the borders
    assert np.array_equal(ctx.map_cache(m, res), oracle.map_cache(m.copy(), res))
    empty = np.zeros((rows, cols), np.uint8)                                 # no occupied cell at all
    assert np.array_equal(ctx.map_cache(empty, res), np.full((rows, cols), 1.0))
    full = np.ones((rows, cols), np.uint8)                                   # every cell occupied
    assert not ctx.map_cache(full, res).any()


def test_map_cache_device_batch(maps, lsdmod, ctx, oracle):
    import torch
    src = maps["aisle2"][:512, :1024]
    batch = np.stack([src, src[::-1].copy(), np.roll(src, 100, 1)]).copy()
    d = torch.from_numpy(batch).cuda()
    out = torch.zeros(batch.shape, dtype=torch.float64, device="cuda")
    ctx.enqueue_map_cache_device(d.data_ptr(), 3, 1024, 512, 0.025, 1.0, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for i in range(3):
        assert np.array_equal(out[i].cpu().numpy(), oracle.map_cache(batch[i].copy(), 0.025))
    assert np.array_equal(d.cpu().numpy(), batch)


# ---- error behaviour of the C ABI (the reference has none: it would crash) -----------------------------------
def test_error_codes(maps, lsdmod, ctx):
    import ctypes
    L, h = ctx.L, ctx.h
    img = maps["map1"].copy()
    rows, cols = img.shape
    P = lsdmod.make_params()
    lines = ctypes.c_void_p()
    n = ctypes.c_int(-1)
    call = lambda im, c_, r_, st, pp: L.lsd_run(h, im, c_, r_, st, ctypes.byref(pp), None, 0, ctypes.byref(lines), ctypes.byref(n))
    assert call(None, cols, rows, cols, P) == lsdmod.LSD_ERR_INVALID
    assert call(img.ctypes.data, cols, rows, cols - 1, P) == lsdmod.LSD_ERR_INVALID          # stride < cols
    assert call(img.ctypes.data, 0, rows, cols, P) == lsdmod.LSD_ERR_INVALID
    assert call(img.ctypes.data, 3, 3, 3, P) == lsdmod.LSD_ERR_INVALID                        # scaled size below 2x2
    assert call(img.ctypes.data, 70000, 4, 70000, P) == lsdmod.LSD_ERR_UNSUPPORTED            # coordinates are packed in 16 bits
    for field, val, want in (("pseBin", 2048, lsdmod.LSD_ERR_UNSUPPORTED), ("pseBin", 0, lsdmod.LSD_ERR_INVALID),
                             ("sca", 0.0, lsdmod.LSD_ERR_INVALID), ("sig", -1.0, lsdmod.LSD_ERR_INVALID)):
        Q = lsdmod.make_params()
        setattr(Q, field, val)
        assert call(img.ctypes.data, cols, rows, cols, Q) == want, field
    assert np.array_equal(img, maps["map1"])                                      # a refused call leaves the map untouched
    out = np.zeros((rows, cols))
    assert L.lsd_map_cache(h, img.ctypes.data, cols, rows, cols, 0.0, 1.0, out.ctypes.data) == lsdmod.LSD_ERR_INVALID
    assert L.lsd_map_cache(h, None, cols, rows, cols, 0.05, 1.0, out.ctypes.data) == lsdmod.LSD_ERR_INVALID
    assert L.lsd_strerror(lsdmod.LSD_ERR_UNSUPPORTED) and L.lsd_last_error(h) is not None
    got_lines, _ = ctx.run(img)                                                   # and the context is still usable
    assert len(got_lines) == 7


def test_scaled_size_limit_and_host_line_capacity(maps, lsdmod, ctx):
    """The region stage keeps scaled coordinates and boxes in 16-bit fields: a scaled width beyond 32766 is refused, not
    mis-computed (sca >= 1 on a wide map).  The host entry points report an image with more lines than their capacity."""
    wide = np.zeros((4, 40000), np.uint8)
    with pytest.raises(lsdmod.LsdError) as e:
        ctx.run(wide, lsdmod.make_params(sca=1.0))
    assert e.value.status == lsdmod.LSD_ERR_UNSUPPORTED
    ctx.set_host_max_lines(8)
    try:
        with pytest.raises(lsdmod.LsdError) as e:
            ctx.run(maps["aisle1"].copy())                                        # 75 lines
        assert e.value.status == lsdmod.LSD_ERR_CAPACITY
    finally:
        ctx.set_host_max_lines(8192)
    lines, _ = ctx.run(maps["aisle1"].copy())
    assert len(lines) == 75


# ---- wire format: OccupancyGrid cells -> map values on the device (byte-exact) ------------
@pytest.mark.parametrize("rows,cols", [(16, 16), (1, 1), (3, 5), (257, 129), (480, 608)])
def test_occupancy_to_map(rows, cols, lsdmod, ctx, oracle):
    rng = np.random.default_rng(rows * 1000 + cols)
    grid = rng.choice(np.array([-1, 0, 100, 1, 37, -128, 127, -2], np.int8), size=(rows, cols),
                      p=[0.3, 0.4, 0.2, 0.02, 0.02, 0.02, 0.02, 0.02]).astype(np.int8)
    if rows * cols >= 256:
        grid.ravel()[:256] = np.arange(256, dtype=np.uint8).view(np.int8)     # every cell value at least once
    assert np.array_equal(ctx.occupancy_to_map(grid), oracle.occupancy_to_map(grid))


def test_occupancy_device_and_map_callback(maps, maps_meta, known, lsdmod, ctx, oracle):
    import torch
    # the grid whose callback output is the aisle1 fixture: 0 -> unknown(-1), 255 -> free(0), 1 -> occupied(100)
    m = maps["aisle1"]
    grid = np.full(m.shape, 100, np.int8); grid[m == 0] = -1; grid[m == 255] = 0
    assert np.array_equal(oracle.occupancy_to_map(grid), m)
    d = torch.from_numpy(np.stack([grid, grid[::-1].copy()])).cuda()
    out = torch.zeros(d.shape, dtype=torch.uint8, device="cuda")
    ctx.enqueue_occupancy_to_map_device(d.data_ptr(), d.numel(), out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(out[0].cpu().numpy(), m) and np.array_equal(out[1].cpu().numpy(), m[::-1])
    # the whole callback: same lines as the fixture run directly, mapCache with the callback's 2 m cap
    rows, cols = m.shape
    res = maps_meta["aisle1"]["res"]
    mapValue, mapCache, LSD = lsdmod.mapCallback(grid.ravel(), cols, rows, res, ctx=ctx)
    assert LSD.len_linesInfo == known["counts"]["aisle1"][0]
    ref = oracle.lsd(m.copy())
    assert np.array_equal(mapValue, ref["map"]) if "map" in ref else True
    assert np.array_equal(mapCache, oracle.map_cache(m.copy(), res, 2.0))


# ---- scan-to-map matching batch: floating point, tolerance stated below -------------------
MATCH_TOL = 1e-9     # scores (metres) and poses (pixels / degrees): differences come only from 1-ulp libm differences in sin/cos/atan


@pytest.mark.parametrize("name,theta,last", [("aisle1", 17.0, (-1.0, -1.0, 0.0)), ("aisle2", -63.0, (-1.0, -1.0, 0.0)),
                                             ("aisle1", 17.0, (300.0, 260.0, 0.0)), ("aisle3", 90.0, (5000.0, 5000.0, 0.0))])
def test_scan_to_map_match_parity(name, theta, last, maps, maps_meta, lsdmod, ctx, oracle):
    from matching_case import build_case
    case = build_case(maps[name], maps_meta[name]["res"], oracle, theta_deg=theta)
    pairs = lsdmod.match_pairs(case["map_lines"], case["scan_lines"])
    assert len(pairs) > 20
    args = (case["map_cache"], case["map_lines"], case["scan_lines"], case["pts"], case["lidar"], last, pairs)
    want = oracle.scan_to_map_match(*args).reshape(-1, 4)
    got = ctx.scan_to_map_match(*args).reshape(-1)
    g = np.stack([got["x"], got["y"], got["ang"], got["score"]], 1)
    assert np.array_equal(np.isinf(g[:, 3]), np.isinf(want[:, 3]))            # the same candidates are rejected
    fin = np.isfinite(want[:, 3])
    if last[0] == 5000.0:
        assert not fin.any()
    else:
        assert fin.sum() > 10
    assert np.allclose(g[fin], want[fin], rtol=0, atol=MATCH_TOL)
    assert np.allclose(g[~fin, :2], want[~fin, :2], rtol=0, atol=MATCH_TOL)   # the rotated lidar position is always reported
    # the host mirror of FeatureAssociation's selection: score < 3, ascending
    if last[0] == -1.0:
        kept = lsdmod.ScanToMapMatch(case["map_cache"], case["map_lines"], case["scan_lines"], case["pts"], case["lidar"], last, ctx=ctx)
        assert len(kept) == int((want[:, 3] < 3).sum()) and np.all(np.diff(kept["score"]) >= 0)
        assert abs(kept["ang"][0] - theta) < 2.0 and np.hypot(kept["x"][0] - case["lidar_map"][0], kept["y"][0] - case["lidar_map"][1]) < 3.0


def test_scan_to_map_match_device_resident(maps, maps_meta, lsdmod, ctx, oracle):
    """mapCache produced on the device feeds the matching kernel directly (nothing but the scan goes over PCIe)."""
    import ctypes, torch
    from matching_case import build_case
    m = maps["aisle1"]; res = maps_meta["aisle1"]["res"]
    case = build_case(m, res, oracle)
    pairs = lsdmod.match_pairs(case["map_lines"], case["scan_lines"])
    rows, cols = m.shape
    d_map = torch.from_numpy(m.copy()).cuda()
    d_mc = torch.zeros((rows, cols), dtype=torch.float64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    ctx.enqueue_map_cache_device(d_map.data_ptr(), 1, cols, rows, res, 1.0, d_mc.data_ptr(), stream=s)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).cuda()
    d_ml, d_sl, d_pts, d_pr = dev(case["map_lines"]), dev(case["scan_lines"]), dev(case["pts"]), dev(pairs)
    d_out = torch.zeros((len(pairs) * 4, 4), dtype=torch.float64, device="cuda")
    P = lsdmod.lsd_position
    ctx._chk(ctx.L.lsd_enqueue_scan_to_map_match_device(ctx.h, d_mc.data_ptr(), cols, rows, d_ml.data_ptr(), d_sl.data_ptr(), d_pts.data_ptr(),
                                                        len(case["pts"]), P(*case["lidar"]), P(-1.0, -1.0, 0.0), d_pr.data_ptr(), len(pairs),
                                                        1.0, 60.0, d_out.data_ptr(), s))
    torch.cuda.synchronize()
    want = oracle.scan_to_map_match(case["map_cache"], case["map_lines"], case["scan_lines"], case["pts"], case["lidar"], (-1.0, -1.0, 0.0), pairs).reshape(-1, 4)
    g = d_out.cpu().numpy()
    fin = np.isfinite(want[:, 3])
    assert np.array_equal(np.isinf(g[:, 3]), ~fin) and np.allclose(g[fin], want[fin], rtol=0, atol=MATCH_TOL)


def test_scan_to_map_match_empty_scan(maps, maps_meta, lsdmod, ctx, oracle):
    """A scan without image points: every candidate is rejected with an infinite score (myFA.cpp:248-263), poses as usual."""
    from matching_case import build_case
    case = build_case(maps["aisle1"], maps_meta["aisle1"]["res"], oracle)
    pairs = lsdmod.match_pairs(case["map_lines"], case["scan_lines"])[:40]
    args = (case["map_cache"], case["map_lines"], case["scan_lines"], np.zeros((0, 3)), case["lidar"], (-1.0, -1.0, 0.0), pairs)
    want = oracle.scan_to_map_match(*args).reshape(-1, 4)
    got = ctx.scan_to_map_match(*args).reshape(-1)
    assert np.all(np.isinf(want[:, 3])) and np.all(np.isinf(got["score"]))
    assert np.allclose(np.stack([got["x"], got["y"], got["ang"]], 1), want[:, :3], rtol=0, atol=MATCH_TOL)


def test_default_variant_for_long_batches(maps, lsdmod, ctx, oracle):
    """More than four images per compute unit: the library runs the region stage with 4 wavefronts per image WITHOUT being told
    to (lsd_ctx.hip: waves_for).  1100 maps of 608 x 480 cut out of the fixtures; 32 of them against the oracle, all of them
    against their duplicates (the crops repeat every 44 images)."""
    srcs = [maps[k] for k in ("map1", "aisle1", "aisle2", "aisle3", "mapValue", "f3key", "f4key")]
    rng = np.random.default_rng(5)
    base = []
    for j in range(44):
        m = srcs[j % len(srcs)]
        if m.shape[0] < 480 or m.shape[1] < 608:
            m = np.pad(m, ((0, max(0, 480 - m.shape[0])), (0, max(0, 608 - m.shape[1]))))
        y0 = int(rng.integers(0, m.shape[0] - 480 + 1)); x0 = int(rng.integers(0, m.shape[1] - 608 + 1))
        base.append(np.ascontiguousarray(m[y0:y0 + 480, x0:x0 + 608]))
    n = 1100
    batch = np.stack([base[i % 44] for i in range(n)])
    ctx.set_region_waves(0)
    lines, offs, ims = ctx.run_batch(batch.copy())
    assert len(offs) == n + 1
    for i in range(32):
        ref = oracle.lsd(base[i].copy())
        assert offs[i + 1] - offs[i] == len(ref["lines"])
        assert np.array_equal(ims[i], ref["lineIm"])
        assert_lines_close(lines[offs[i]:offs[i + 1]], ref["lines"])
    for i in range(44, n):
        j = i % 44
        assert offs[i + 1] - offs[i] == offs[j + 1] - offs[j]
        assert lines[offs[i]:offs[i + 1]].tobytes() == lines[offs[j]:offs[j + 1]].tobytes()


def test_certified_uniform_sets_answer_like_full_evaluations(maps, lsdmod, ctx, oracle):
    """The heaviest bench images are one sparse structure of pixels with the SAME level-line angle, grown again from each of its
    hundreds of seeds and given up by Refiner every time (k_region.hip, "Certified uniform sets").  The region stage founds a set from
    the first such evaluation and answers the structure's other seeds without growing anything.  Every seed's record -- first region,
    final region, outcome, logNFA -- must be the oracle's, with the trace on (every record through the cursor one by one) and off, and
    most of the structure's seeds must have been answered by the set."""
    import bench
    for i in (187, 16):
        img = bench.make_image(maps, i, 2048)
        ref = oracle.lsd(img.copy(), debug=True)
        d = ref["dbg"]
        rs = d["seeds"]
        big1 = int(((rs["outcome"] == 1) & (rs["num"] >= 256)).sum())
        assert big1 > 100
        ctx.set_trace(True)
        try:
            ctx.run(img.copy(), want_lineim=False)
            seeds = ctx.fetch(0, lsdmod.DBG_SEEDS, (d["w"], d["h"]))
            st = ctx.fetch(0, lsdmod.DBG_STATS, (d["w"], d["h"]))
        finally:
            ctx.set_trace(False)
        assert len(seeds) == len(rs)
        for f in ("order_idx", "x", "y", "num", "outcome", "final_num"):
            assert np.array_equal(seeds[f], rs[f]), (i, f)
        assert st["sets_founded"] >= 1 and st["set_answers"] >= big1 // 2, (i, st["sets_founded"], st["set_answers"], big1)
        for waves in (4, 8):
            ctx.set_region_waves(waves)
            try:
                lines, im = ctx.run(img.copy())
            finally:
                ctx.set_region_waves(0)
            used = (ctx.fetch(0, lsdmod.DBG_STATE, (d["w"], d["h"])) & 3).astype(np.uint8)
            assert np.array_equal(used, d["used"]) and np.array_equal(im, ref["lineIm"])
            assert_lines_close(lines, ref["lines"])


def test_nfa_values_equal_the_correctly_rounded_restatement(maps, lsdmod, ctx, oracle):
    """RectangleNFACalculator's libm calls on the device -- exp of the first term, log10 of the tail, pow / log10 of the tail's stopping
    test -- are correctly rounded (crmath.h; the stopping test from the device math library's values where a bracket decides it): on
    every fixture and on bench images every seed's logNFA is the one of the restatement built on correctly rounded functions
    (oracle/liblsd_oracle_cr.so), bit for bit; and the bracket leaves few stopping tests to the slow path."""
    import bench
    imgs = [maps[k] for k in FIXTURES] + [bench.make_image(maps, i, 2048) for i in (0, 187)]
    ctx.set_trace(True)
    try:
        for im in imgs:
            d = oracle.lsd(im.copy(), debug=True, _lib=oracle.lib_cr())["dbg"]
            ctx.run(im.copy(), want_lineim=False)
            seeds = ctx.fetch(0, lsdmod.DBG_SEEDS, (d["w"], d["h"]))
            st = ctx.fetch(0, lsdmod.DBG_STATS, (d["w"], d["h"]))
            assert len(seeds) == len(d["seeds"])
            for f in ("order_idx", "num", "outcome", "final_num"):
                assert np.array_equal(seeds[f], d["seeds"][f]), f
            assert np.array_equal(seeds["logNFA"], d["seeds"]["logNFA"])
            assert st["nfa_calls"] > 0 and st["nfa_bracket_misses"] <= st["nfa_calls"] // 100
    finally:
        ctx.set_trace(False)


def test_device_libm_is_inside_the_nfa_bracket(lsdmod, ctx):
    """The stopping test of the binomial tail (myLSD.cpp:1052-1053) is decided from the device math library's pow and log10 wherever
    a bracket of 2^-44 relative (kOcmlBracket in k_region.hip) around them decides it: the library's values must lie within a
    sixteenth of that bracket of the correctly rounded ones, over the arguments the NFA produces (a ratio in (0, 1) to a pixel count;
    tails between 1e-300 and 1e3)."""
    rng = np.random.default_rng(7)
    n = 400000
    x = np.concatenate([rng.uniform(0, 1, n), 1 - 10.0 ** rng.uniform(-12, 0, n), 10.0 ** rng.uniform(-8, 0, n)])
    y = np.concatenate([rng.integers(1, 6000, n), rng.integers(1, 400000, n), rng.integers(1, 50, n)]).astype(np.float64)
    cr, dev = ctx.eval_math(6, x, y)
    ok = cr > 1e-290
    assert np.all(np.abs(dev[ok] - cr[ok]) <= 2.0 ** -48 * cr[ok]) and np.all(dev[~ok] <= 1e-289)
    t = np.concatenate([10.0 ** rng.uniform(-300, 3, n), 1 + rng.normal(0, 1, n) * 10.0 ** rng.uniform(-15, -1, n)])
    t = np.abs(t[t > 0])
    cr, dev = ctx.eval_math(5, t)
    assert np.all(np.abs(dev - cr) <= 2.0 ** -48 * np.abs(cr) + 1e-300)


def test_nfa_decisions_are_far_from_ties(maps, lsdmod, ctx):
    """RectangleImprover's decisions (logNFA > 0, candidate > best so far) are those of correctly rounded arithmetic (above); glibc's
    exp / log10 / pow are not correctly rounded (test_crmath.py: log10 off by one ulp in one call out of seven), so a decision of the
    reference could differ where two compared values sit within what those last places can move them.  The region stage records every
    comparison's MARGIN -- the distance of its operands over that noise (k_region.hip: improve()); below 1 a flip is possible.  What
    comes closest are STRUCTURAL near-ties: the binomial tail B(1/p + 1, 1/p, p) equals p^(1/p - 1) = B(1/p - 1, 1/p - 1, p) exactly, and
    the two evaluations (the sum through the Lanczos / Windschitl log-gamma, the closed form) differ by the log-gamma formulas' own
    error, ~3e-13 absolute -- the same on every libm.  On every fixture and on the whole 512-image bench batch every margin is >= 10."""
    import torch
    import bench
    lo_abs, lo_gap, calls = float("inf"), float("inf"), 0
    for im in [maps[k] for k in FIXTURES]:
        ctx.run(im.copy(), want_lineim=False)
        st = ctx.fetch(0, lsdmod.DBG_STATS, lsdmod.scaled_size(im.shape[1], im.shape[0]))
        assert st["nfa_calls"] > 0 and np.isfinite(st["nfa_min_abs"])
        lo_abs, lo_gap, calls = min(lo_abs, st["nfa_min_abs"]), min(lo_gap, st["nfa_min_gap"]), calls + st["nfa_calls"]
    n, size = 512, 2048
    d = torch.from_numpy(bench.make_batch(maps, n, size)).cuda()
    d_lines = torch.zeros((n, 1024, 10), dtype=torch.int64, device="cuda")
    d_counts = torch.zeros(n, dtype=torch.int32, device="cuda")
    ctx.enqueue_device(d.data_ptr(), n, size, size, d_lines.data_ptr(), 1024, d_counts.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for i in range(n):
        st = ctx.fetch(i, lsdmod.DBG_STATS, lsdmod.scaled_size(size, size))
        lo_abs, lo_gap, calls = min(lo_abs, st["nfa_min_abs"]), min(lo_gap, st["nfa_min_gap"]), calls + st["nfa_calls"]
    del d, d_lines
    torch.cuda.empty_cache()
    assert calls > 1_000_000
    assert lo_abs >= 10, lo_abs
    assert lo_gap >= 10, lo_gap


def test_rccl_gather_runs_on_the_gpu(maps, lsdmod, ctx):
    """The multi-GPU hand-off (dist.gather_line_lists over backend nccl = RCCL) on the one GPU there is: bench.py as a fresh child
    process under torch.distributed.run with one rank and --force-dist (the launcher comes first in that child, before
    anything touches the GPU).  bench.py itself asserts that the gathered counts add up to the lines of the step and that no
    slab overflowed; here the step's line total is checked against run_batch on the same six images."""
    import json, socket, subprocess
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--batch", "6", "--size", "1024",
           "--steps", "1", "--warmup", "1", "--no-cpu-baseline"]
    p = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["config"]["images_total"] == 6
    lines, offs, _ = ctx.run_batch(bench.make_batch(maps, 6, 1024))
    assert int(out["lines_per_step"]) == len(lines) == int(offs[-1])
    # the strong split timed next to the weak job (with one rank: the same six images as one shard) went through as well, and the
    # gathered copies bench.py asserts afterwards are the weak job's again
    ss = out["strong_split"]
    assert ss and ss["scaling"] == "strong" and ss["images_total"] == 6 and ss["ms_per_step"] > 0


def _fnv1a(b):
    h = 1469598103934665603
    for x in b:
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_c_abi_hand_off_over_rccl_from_a_cpp_host(maps, lsdmod, ctx, tmp_path):
    """The multi-GPU hand-off as a C++ host calls it (include/lsd_hip.h: lsd_shard_range, lsd_comm_from_rccl, lsd_gather_lines,
    lsd_gather_unpack; include/myLSD.h: set_device / context), in a fresh child process, over a real RCCL communicator of world
    size 1 (tests/gather_host.cpp): offsets and the bytes of the line records equal those of run_batch on the same images."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "linesegmentdetector-slam_amd")
    batch = bench.make_batch(maps, 5, 1024, 8)
    raw = tmp_path / "batch.raw"
    batch.tofile(raw)
    exe = tmp_path / "gather_host"
    subprocess.run(["g++", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(root, "include"), "-I", "/opt/rocm/include",
                    os.path.join(root, "tests", "gather_host.cpp"), "-o", str(exe), "-L", pkg, "-llsdhip", "-L/opt/rocm/lib", "-lamdhip64", "-lrccl",
                    "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([str(exe), str(raw), "5", "1024", "1024"], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-2000:])
    tok = [l for l in p.stdout.splitlines() if l.startswith("world")][-1].split()
    assert tok[:6] == ["world", "1", "rank", "0", "per", "5"]
    offs_got = [int(v) for v in tok[7:13]]
    lines, offs, _ = ctx.run_batch(batch.copy(), want_lineim=False)
    assert offs_got == offs.tolist() and offs_got[-1] > 100
    assert int(tok[14]) == _fnv1a(lines.tobytes())


def test_c_abi_hand_off_world2_on_one_gpu(maps, lsdmod, ctx):
    """lsd_gather_lines with TWO ranks on the one GPU there is: two contexts, each running its lsd_shard_range shard of a 7-image
    batch, and a communicator whose all_gather callback copies between the two ranks' device buffers (the same lsd_comm the RCCL
    binding fills).  Both ranks end up with the same gathered arrays; unpacked, they are run_batch's answer for the whole batch,
    in global image order.  A slab too small for a rank's lines is flagged on every rank."""
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    ldist = importlib.import_module("linesegmentdetector-slam_amd.dist")
    n_total, size, world, max_lines = 7, 1024, 2, 256
    batch = bench.make_batch(maps, n_total, size, 40)
    ref_lines, ref_offs, _ = ctx.run_batch(batch.copy(), want_lineim=False)
    per, words = lsdmod.gather_layout(n_total, world)
    ctxs = [lsdmod.Context(0) for _ in range(world)]
    stream = torch.cuda.current_stream().cuda_stream
    try:
        for cap_rows, expect_over in ((n_total * 128, False), (100, True)):
            calls = []                                           # (rank, collective number, d_send, d_recv, bytes)
            def make_comm(r):
                k = [0]
                def all_gather(user, d_send, d_recv, nbytes, st):
                    calls.append((r, k[0], d_send, d_recv, nbytes)); k[0] += 1
                    return 0
                cb = lsdmod.ALL_GATHER_FN(all_gather)
                comm = lsdmod.lsd_comm(r, world, cb, None)
                comm._keep = cb
                return comm
            outs = []
            for r in range(world):
                lo, hi = lsdmod.shard_range(n_total, world, r)
                d = torch.from_numpy(batch[lo:hi]).cuda()
                L = torch.zeros((hi - lo, max_lines, 10), dtype=torch.int64, device="cuda"); cnt = torch.zeros(hi - lo, dtype=torch.int32, device="cuda")
                ctxs[r].enqueue_device(d.data_ptr(), hi - lo, size, size, L.data_ptr(), max_lines, cnt.data_ptr(), stream=stream)
                ca = torch.zeros((world, per + 2), dtype=torch.int32, device="cuda"); sl = torch.zeros((world, cap_rows, 10), dtype=torch.int64, device="cuda")
                ctxs[r].gather_lines(make_comm(r), L.data_ptr(), cnt.data_ptr(), hi - lo, max_lines, n_total, cap_rows, ca.data_ptr(), sl.data_ptr(), stream)
                outs.append((ca, sl, d, L, cnt))
            torch.cuda.synchronize()
            # the "collective": rank q's send buffer of collective k lands at slot q of every rank's receive buffer
            assert len(calls) == 2 * world
            for (r, k, _, d_recv, nb) in calls:
                for (q, k2, d_send, _, nb2) in calls:
                    if k2 == k:
                        assert nb2 == nb
                        dst = torch.as_tensor(ldist._RawDeviceBytes(d_recv + q * nb, nb), device="cuda")
                        dst.copy_(torch.as_tensor(ldist._RawDeviceBytes(d_send, nb), device="cuda"))
            torch.cuda.synchronize()
            assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
            ca, sl = outs[1][0].cpu().numpy(), outs[1][1].cpu().numpy()
            if expect_over:
                assert ca[:, per + 1].any()
                with pytest.raises(lsdmod.LsdError) as e:
                    lsdmod.gather_unpack(ca, sl, n_total, world, cap_rows)
                assert e.value.status == lsdmod.LSD_ERR_CAPACITY
            else:
                offs, lines = lsdmod.gather_unpack(ca, sl, n_total, world, cap_rows)
                assert np.array_equal(offs, ref_offs) and lines.tobytes() == ref_lines.tobytes()
        with pytest.raises(lsdmod.LsdError):                     # a shard that is not this rank's
            ctxs[0].gather_lines(make_comm(0), outs[1][3].data_ptr(), outs[1][4].data_ptr(), outs[1][4].numel(), max_lines, n_total, 64,
                                 outs[0][0].data_ptr(), outs[0][1].data_ptr(), stream)
    finally:
        for c in ctxs: c.close()


def test_feature_scan_batch_matches_oracle(lsdmod, ctx, oracle):
    """myrdp::FeatureScan for all 99 frames of data/Lidar.txt in ONE launch against the C restatement, frame by frame: line
    records (integers and the correctly rounded atand / cosd / sind: k, b, x, y, len, orient exact, dx / dy to 1e-15), the list of
    raster pixels in order, lidarPos and the image size; and the MATLAB golden of frame 31 through the C ABI."""
    from test_oracle import RDP_MAP_PARAM, rdp_golden_check
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "lidar.npz"))
    lid = z["lidar"]
    n = len(lid)
    scans = np.zeros((n, 360, 2)); lens = np.zeros(n, np.int32)
    for f in range(n):
        sc = lid[f][np.isfinite(lid[f][:, 0])]
        scans[f, :len(sc)] = sc; lens[f] = len(sc)
    got = ctx.feature_scan_batch(scans, lens, RDP_MAP_PARAM)
    for f in range(n):
        ref = oracle.feature_scan(scans[f, :lens[f]], RDP_MAP_PARAM)
        g = got[f]
        assert g["len_linesInfo"] == len(ref["lines"]), f
        assert g["lidarPos"] == ref["lidar_pos"] and g["lineIm"].shape == ref["lineIm"].shape
        for k in ("x1", "y1", "x2", "y2", "orient"):
            assert np.array_equal(g["linesInfo"][k], ref["lines"][k]), (f, k)
        for k in ("k", "b", "len"):
            assert np.array_equal(g["linesInfo"][k], ref["lines"][k], equal_nan=True), (f, k)
        for k in ("dx", "dy"):
            assert np.abs(g["linesInfo"][k] - ref["lines"][k]).max() <= 1e-15, (f, k)
        assert np.array_equal(g["scanImPoint"], ref["pts"]), f
        assert np.array_equal(g["lineIm"], ref["lineIm"])
    assert rdp_golden_check(got[int(z["matlab_frame"])], z) == (8, 711)
    one = lsdmod.FeatureScan(RDP_MAP_PARAM, scans[5, :lens[5]], ctx=ctx)
    assert np.array_equal(one["scanImPoint"], got[5]["scanImPoint"]) and one["linesInfo"].tobytes() == got[5]["linesInfo"].tobytes()


def test_helper_pool_on_single_images_is_repeatable(maps, lsdmod, oracle):
    """A call with a handful of images gets helper-only workgroups from the start of the region stage (lsd_ctx.hip: pool_for): the
    heaviest bench image and two fixtures, each alone and 12 times over, must give the oracle's answer every time -- byte for byte the
    answer of a context with the help switched off -- and the heavy image must actually have been helped."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    cases = [("bench187", bench.make_image(maps, 187, 2048)), ("f4key", maps["f4key"]), ("f3key", maps["f3key"])]
    on, off = lsdmod.Context(0), lsdmod.Context(0)
    on.set_region_help(24)                                                    # (off by default since round 6: the pool has to be asked for)
    off.set_region_help(0)
    try:
        for name, img in cases:
            ref = oracle.lsd(img.copy())
            l0, im0 = off.run(img.copy())
            assert_lines_close(l0, ref["lines"]); assert np.array_equal(im0, ref["lineIm"])
            helped = 0
            for rep in range(12):
                l1, im1 = on.run(img.copy())
                assert l1.tobytes() == l0.tobytes() and np.array_equal(im1, im0), (name, rep)
                helped += on.fetch(0, lsdmod.DBG_STATS, lsdmod.scaled_size(img.shape[1], img.shape[0]))["help_exports"]
            if name == "bench187":
                assert helped > 1000, helped
    finally:
        on.close(); off.close()


def test_feature_scan_more_lines_than_the_reference_array_holds(lsdmod, ctx, oracle):
    """A 1024-reading zigzag whose every reading becomes a split point gives ~1000 chords; the reference's array holds 360 (:39).
    n_lines reports them all, the first 360 records are stored and equal the oracle's, the host entry point says LSD_ERR_CAPACITY."""
    import ctypes as C
    from test_oracle import RDP_MAP_PARAM
    n = 1024
    ang = np.linspace(-2.0, 2.0, n)
    scan = np.stack([5.0 + 0.3 * (-1.0) ** np.arange(n), ang], 1)
    ref = oracle.feature_scan(scan, RDP_MAP_PARAM, 3, 0.01, 0.0, pts_cap=65536)
    assert ref["n_lines"] > 360 and len(ref["lines"]) == 360
    with pytest.raises(lsdmod.LsdError) as e:
        ctx.feature_scan_batch(scan[None], [n], RDP_MAP_PARAM, 3, 0.01, 0.0, pts_cap=65536)
    assert e.value.status == lsdmod.LSD_ERR_CAPACITY
    lines = np.zeros((1, 360), lsdmod.LINE_DTYPE); pts = np.zeros((1, 65536, 3)); nl = np.zeros(1, np.int32); npt = np.zeros(1, np.int32)
    lp = np.zeros((1, 2)); sz = np.zeros((1, 2), np.int32); ln = np.array([n], np.int32)
    mp = lsdmod.lsd_map_param(int(RDP_MAP_PARAM[0]), int(RDP_MAP_PARAM[1]), *[float(v) for v in RDP_MAP_PARAM[2:]])
    st = ctx.L.lsd_feature_scan_batch(ctx.h, np.ascontiguousarray(scan).ctypes.data, ln.ctypes.data, 1, n, mp, 3, 0.01, 0.0, lines.ctypes.data, nl.ctypes.data,
                                      pts.ctypes.data, 65536, npt.ctypes.data, lp.ctypes.data, sz.ctypes.data)
    assert st == lsdmod.LSD_ERR_CAPACITY and nl[0] == ref["n_lines"] and npt[0] == len(ref["pts"])
    lines["_pad"] = 0
    assert lines[0].tobytes() == ref["lines"].tobytes()
    assert np.array_equal(pts[0, :npt[0]], ref["pts"])


@pytest.mark.parametrize("waves", [8, 4])
def test_help_across_workgroups_changes_nothing(waves, maps, lsdmod, oracle):
    """One of the heaviest bench images in a launch with 23 copies of a light one: the workgroups of the light images finish
    early and their wavefronts evaluate seeds of the heavy image (k_region.hip, "Help from other workgroups").  The counters say
    that they did; lines, line records and lineIm are those of the oracle and, byte for byte, those of a context with the help
    switched off (LSD_REGION_HELP=0, read when a context is created)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    heavy, light = bench.make_image(maps, 187, 2048), bench.make_image(maps, 0, 2048)
    batch = np.stack([heavy] + [light] * 23)
    wh = lsdmod.scaled_size(2048, 2048)
    res = {}
    for help_ in ("24", "0"):
        old = os.environ.get("LSD_REGION_HELP")
        os.environ["LSD_REGION_HELP"] = help_
        try:
            c = lsdmod.Context(0)
        finally:
            if old is None: del os.environ["LSD_REGION_HELP"]
            else: os.environ["LSD_REGION_HELP"] = old
        c.set_region_waves(waves)
        lines, offs, ims = c.run_batch(batch.copy())
        st = [c.fetch(i, lsdmod.DBG_STATS, wh) for i in range(len(batch))]
        res[help_] = (lines.tobytes(), offs.tobytes(), ims.tobytes(), st[0]["help_exports"], sum(x["help_evals"] for x in st))
        if help_ == "24":
            for j in (0, 1):
                ref = oracle.lsd(batch[j].copy())
                assert offs[j + 1] - offs[j] == len(ref["lines"])
                assert_lines_close(lines[offs[j]:offs[j + 1]], ref["lines"])
                assert np.array_equal(ims[j], ref["lineIm"])
        c.close()
    assert res["24"][:3] == res["0"][:3]
    assert res["0"][3] == 0 and res["0"][4] == 0
    assert res["24"][3] > 20 and res["24"][4] > 20, res["24"][3:]      # (hundreds on an idle device)


def test_large_batches_run_without_help_by_default(maps, lsdmod, oracle):
    """Help across workgroups is off unless lsd_set_region_help asks for it (since round 6 at every call size): no image exports a
    seed by default -- and asking changes no byte of the result."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    heavy = bench.make_image(maps, 27, 2048)[:1024, :1024]
    light = bench.make_image(maps, 0, 2048)[:1024, :1024]
    batch = np.ascontiguousarray(np.stack([heavy] + [light] * 65))
    wh = lsdmod.scaled_size(1024, 1024)
    res = {}
    for help_ in (-1, 24):
        c = lsdmod.Context(0)
        c.set_region_waves(8); c.set_region_help(help_)
        lines, offs, ims = c.run_batch(batch.copy())
        res[help_] = (lines.tobytes(), offs.tobytes(), ims.tobytes(), sum(c.fetch(i, lsdmod.DBG_STATS, wh)["help_exports"] for i in range(len(batch))))
        c.close()
    assert res[-1][3] == 0, res[-1][3]
    assert res[-1][:3] == res[24][:3]
    ref = oracle.lsd(batch[0].copy())
    n0 = np.frombuffer(res[-1][1], np.int32 if len(res[-1][1]) == 4 * (len(batch) + 1) else np.int64)
    assert n0[1] - n0[0] == len(ref["lines"])


@pytest.mark.parametrize("tun", [{"REQUEUE": 0}, {"SOFT": 64, "CLAIM": 64}, {"SOFT": 1900, "CLAIM": 1900, "BIG": 16},
                                 {"HELP": 64, "WB": 100, "XPOLL": 2000},
                                 # the 8-wave region stage as 5 / 47 PERSISTENT workgroups that share the 48 images out through a counter
                                 # (k_region.hip: k_region; the default takes that path for batches of more images than CUs, help off)
                                 {"HELP": 0, "GROUPS": 5}, {"HELP": 0, "GROUPS": 47}, {"HELP": 0, "GROUPS": 0}])
def test_schedule_of_the_region_stage_changes_nothing(tun, maps, lsdmod, ctx):
    """How far the wavefronts work ahead of the commit cursor, whether invalidated results are re-queued when a line is accepted
    or found at the cursor, who may ask for help and how often the help protocol is looked at: all of it is schedule.  A
    48-image slice of the bench batch (1024 x 1024, heavy and light images mixed) gives the same bytes under each setting as
    under the defaults (lsd_debug_set_tuning; the shipped library takes none of these from the environment)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    batch = bench.make_batch(maps, 48, 1024, 140)
    ref = ctx.run_batch(batch.copy())
    c = lsdmod.Context(0)
    try:
        for k, v in tun.items():
            c.debug_set_tuning(k, v)
        with pytest.raises(lsdmod.LsdError):
            c.debug_set_tuning("NO_SUCH_SETTING", 1)
        for rep in range(2):
            got = c.run_batch(batch.copy())
            assert all(a.tobytes() == b.tobytes() for a, b in zip(got, ref))
    finally:
        c.close()


def test_cost_history_changes_the_order_not_the_result(maps, lsdmod, ctx):
    """lsd_set_cost_history: the region stage starts a batch's images in the order of their cost in the context's previous call.  The same
    96-image batch gives the same bytes with the hint (after a first call that records the costs) as without, and lsd_last_region_cycles
    reports a cost for every image."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    batch = bench.make_batch(maps, 96, 1024, 140)
    ref = ctx.run_batch(batch.copy())
    c = lsdmod.Context(0)
    try:
        c.set_cost_history(True)
        for rep in range(3):
            got = c.run_batch(batch.copy())
            assert all(a.tobytes() == b.tobytes() for a, b in zip(got, ref)), rep
        cyc = c.last_region_cycles(96)
        assert cyc.shape == (96,) and np.all(cyc > 0)
        perm = lsdmod.shard_balanced(cyc, 4)
        assert sorted(perm.tolist()) == list(range(96))
    finally:
        c.close()


def test_batches_in_flight_on_several_contexts(maps, lsdmod, ctx):
    """Throughput mode (bench.py --pipeline): three batches in flight, one context and one stream each,
    help across workgroups off, so that the workgroups of one batch's region stage fill the CUs the other's finished images
    leave idle.  Every batch's counts, line records and lineIm equal those of the same batch run alone."""
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    n, size = 96, 1024
    hosts = [bench.make_batch(maps, n, size, first) for first in (0, 96, 192)]
    def run(c, d, outs, stream):
        c.enqueue_device(d.data_ptr(), n, size, size, outs[0].data_ptr(), 1024, outs[1].data_ptr(), d_line_ims=outs[2].data_ptr(), stream=stream)
    mk = lambda: (torch.zeros((n, 1024, 10), dtype=torch.int64, device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda"),
                  torch.zeros((n, size, size), dtype=torch.uint8, device="cuda"))
    devs = [torch.from_numpy(h).cuda() for h in hosts]
    alone = []
    for d in devs:
        o = mk()
        run(ctx, d, o, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        alone.append([t.cpu().numpy().tobytes() for t in o])
    ctxs = [lsdmod.Context(0) for _ in devs]
    streams = [torch.cuda.Stream() for _ in devs]
    outs = [mk() for _ in devs]
    try:
        for c in ctxs:
            c.set_region_help(0)
            c.reserve(n, size, size)
        for rep in range(3):                               # (the launches of the three streams interleave differently every time)
            for o in outs:
                for t in o: t.zero_()
            torch.cuda.synchronize()
            for c, d, o, st in zip(ctxs, devs, outs, streams):
                run(c, d, o, st.cuda_stream)
            torch.cuda.synchronize()
            for o, ref in zip(outs, alone):
                assert [t.cpu().numpy().tobytes() for t in o] == ref
    finally:
        for c in ctxs: c.close()
    with pytest.raises(lsdmod.LsdError):
        ctx.set_region_help(-2)


def test_region_stage_variants_agree(maps, lsdmod, ctx, oracle):
    """The region stage exists with 4 and with 8 wavefronts per image (chosen by batch size): same lines, same usedMap."""
    crop = lambda a: np.ascontiguousarray(a[:600, :1600])
    imgs = np.stack([crop(maps["aisle1"]), crop(maps["aisle2"]), crop(maps["aisle3"])])
    res = {}
    try:
        for waves in (4, 8):
            ctx.set_region_waves(waves)
            lines, offs, ims = ctx.run_batch(imgs.copy())
            used = [(ctx.fetch(i, lsdmod.DBG_STATE, lsdmod.scaled_size(imgs.shape[2], imgs.shape[1])) & 3).astype(np.uint8) for i in range(3)]
            res[waves] = (lines, offs, ims, used)
    finally:
        ctx.set_region_waves(0)
    a, b = res[4], res[8]
    assert np.array_equal(a[1], b[1]) and a[0].tobytes() == b[0].tobytes() and np.array_equal(a[2], b[2])
    assert all(np.array_equal(x, y) for x, y in zip(a[3], b[3]))
    for i in range(3):
        assert a[1][i + 1] - a[1][i] == len(oracle.lsd(imgs[i].copy())["lines"])


def _sawtooth(name):
    if name == "saw_x":                                                      # bands of ~36 x 450 scaled pixels, one angle each
        yy, xx = np.mgrid[0:1500, 0:900]
        return ((xx % 120) * 255 // 119).astype(np.uint8)
    if name == "saw_diag":
        yy, xx = np.mgrid[0:1200, 0:1000]
        return (((xx + yy) % 170) * 255 // 169).astype(np.uint8)
    if name == "saw_tall":                                                   # more than 65535 pixels per region
        yy, xx = np.mgrid[0:7000, 0:400]
        return ((xx % 120) * 255 // 119).astype(np.uint8)
    rng = np.random.default_rng(4)
    yy, xx = np.mgrid[0:1200, 0:900]
    return np.clip((xx % 120) * 255.0 / 119 + rng.normal(0, 3, xx.shape), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("name", ["saw_x", "saw_diag", "saw_tall", "saw_noise"])
def test_giant_regions_match_oracle(name, lsdmod, ctx, oracle):
    """Occupancy maps have thin walls; a sawtooth image has regions of tens of thousands of pixels instead.  They leave every
    fast structure of the region stage (LDS list -> HBM spill, worklists and skip filter given up, 16-bit worklist indices
    exceeded, result slots too small for the examined list) and must still give the reference's answer, with either build."""
    img = _sawtooth(name)
    try:
        for waves in (4, 8):
            ctx.set_region_waves(waves)
            lines, _, ref = full_check(lsdmod, ctx, oracle, img)
            assert len(lines) > 0 and ref["dbg"]["grown_px"] > 20000
    finally:
        ctx.set_region_waves(0)


def test_stamp_ids_running_out_changes_nothing(maps, lsdmod, ctx, oracle):
    """A wavefront that uses up its 2^20 curMap stamp ids inside one run clears its stamps and starts over; with the budget
    lowered to a few grows every image takes that path hundreds of times."""
    img = np.ascontiguousarray(maps["aisle2"][:700, :1500])
    try:
        for waves, budget in ((4, 7), (8, 3), (8, 64)):
            ctx.set_region_waves(waves)
            ctx.debug_set_stamp_budget(budget)
            full_check(lsdmod, ctx, oracle, img)
    finally:
        ctx.debug_set_stamp_budget(0xFFFF0)
        ctx.set_region_waves(0)


def test_map_cache_many_small_maps_one_workgroup_each(lsdmod, ctx, oracle):
    """More than 64 maps take the one-workgroup-per-map kernel (fewer take the kernel-per-level one): same answers."""
    import torch
    rng = np.random.default_rng(5)
    n, rows, cols = 70, 96, 160
    batch = np.zeros((n, rows, cols), np.uint8)
    batch[rng.random(batch.shape) < 0.01] = 1
    batch[rng.random(batch.shape) < 0.2] = 255
    d = torch.from_numpy(batch).cuda()
    out = torch.zeros(batch.shape, dtype=torch.float64, device="cuda")
    ctx.enqueue_map_cache_device(d.data_ptr(), n, cols, rows, 0.05, 1.0, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for i in (0, 1, 35, 69):
        assert np.array_equal(got[i], oracle.map_cache(batch[i].copy(), 0.05))
    few = torch.zeros((5, rows, cols), dtype=torch.float64, device="cuda")
    ctx.enqueue_map_cache_device(d.data_ptr(), 5, cols, rows, 0.05, 1.0, few.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(few.cpu().numpy(), got[:5])


LIBM_TIES = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "libm_ties.json")))


@pytest.mark.parametrize("name", sorted(LIBM_TIES))
def test_libm_tie_images_equal_the_correctly_rounded_restatement(name, lsdmod, ctx, oracle):
    """ALL disagreements with the glibc-built oracle the random campaigns found (33 of 160 000 images, 2 of 3 900 large ones:
    tests/golden/make_libm_ties.py): on every one the HIP path is bit-identical to the same restatement built on correctly rounded
    sin / cos / atan2 / exp / log10 / pow -- usedMap, lineIm, line records, every seed's decision and NFA value."""
    img = np.load(os.path.join(os.path.dirname(__file__), "golden", "libm_ties.npz"))[name]
    kw = LIBM_TIES[name]["params"]
    ref = oracle.lsd(img.copy(), debug=True, _lib=oracle.lib_cr(), **kw)
    d = ref["dbg"]
    ctx.set_trace(True)
    try:
        lines, im = ctx.run(img.copy(), lsdmod.make_params(**kw) if kw else None)
        seeds = ctx.fetch(0, lsdmod.DBG_SEEDS, (d["w"], d["h"]))
    finally:
        ctx.set_trace(False)
    used = (ctx.fetch(0, lsdmod.DBG_STATE, (d["w"], d["h"])) & 3).astype(np.uint8)
    assert np.array_equal(used, d["used"]) and np.array_equal(im, ref["lineIm"])
    assert_lines_close(lines, ref["lines"])
    assert len(seeds) == len(d["seeds"])
    for f in ("order_idx", "num", "outcome", "final_num"):
        assert np.array_equal(seeds[f], d["seeds"][f]), f
    assert np.array_equal(seeds["logNFA"], d["seeds"]["logNFA"])      # every NFA value, to the bit
    g = oracle.lsd(img.copy(), debug=True, **kw)                      # ... and differs from the glibc-built one (the caveat)
    same = np.array_equal(used, g["dbg"]["used"]) and np.array_equal(im, g["lineIm"])
    assert same == name.startswith("near")                            # (near*: an NFA comparison inside the libms' noise on which both builds agree)
    # ... and the library says so itself: the image has decisions within the noise of the reference's libm (lsd_last_sensitivity);
    # with tracing off too (the count does not depend on it)
    assert int(ctx.last_sensitivity(1)[0]) > 0
    ctx.run(img.copy(), lsdmod.make_params(**kw) if kw else None)
    assert int(ctx.last_sensitivity(1)[0]) > 0 and ctx.fetch(0, lsdmod.DBG_STATS, (d["w"], d["h"]))["near_ties"] > 0


def test_sensitivity_of_the_reference_maps_and_of_a_batch(maps, lsdmod, ctx, oracle):
    """lsd_last_sensitivity on the reference's own maps: the count of decisions within the libm's noise is reported per image of a
    batch, and the call fails after a run that stopped before the region stage."""
    names = [n for n in FIXTURES if maps[n].shape == maps["aisle2"].shape]    # aisle2 / aisle3 share a size
    batch = np.stack([maps[n] for n in names] + [maps[names[0]]])
    ctx.run_batch(batch.copy())
    t = ctx.last_sensitivity(len(batch))
    assert t.dtype == np.int32 and (t >= 0).all()
    with pytest.raises(lsdmod.LsdError):
        ctx.last_sensitivity(len(batch) + 1)
    counts = {}
    for n in FIXTURES:
        ctx.run(maps[n].copy())
        counts[n] = int(ctx.last_sensitivity(1)[0])
    print("near ties of the reference's maps:", counts, "in a batch:", t)
    # (the count includes speculative evaluations that were discarded, so it is an upper bound that moves a little with the schedule;
    #  what does not move is whether an image has any: every one of the reference's maps does -- the END edges of a rectangle pass
    #  through the centre of the region's extreme pixel by construction, :701-720, so "which side of the edge is that pixel on"
    #  is decided in the last places of sin / cos on nearly every map)
    for j, n in enumerate(names):
        assert (counts[n] > 0) == (t[j] > 0), (n, counts[n], t[j])
    try:
        ctx.set_stop_after(lsdmod.STAGE_SORT)
        ctx.run(maps["map1"].copy())
        with pytest.raises(lsdmod.LsdError):
            ctx.last_sensitivity(1)
        with pytest.raises(lsdmod.LsdError):
            ctx.last_region_cycles(1)
    finally:
        ctx.set_stop_after(lsdmod.STAGE_ALL)


def test_watchdog_failure_path_is_reported_not_fatal(maps, lsdmod, ctx):
    """The region stage's watchdog gives an image up with counts = -1.  A test build of the library (make wdtest: odd images never
    publish a full evaluation, the watchdog fires after ~10 ms) drives that path: the device entry point leaves -1 in the counts of
    the odd images and the right answers in the even ones; the host entry point returns LSD_ERR_INTERNAL naming the first such
    image, with offsets and lines that skip the given-up images instead of running out of bounds."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    wd = os.path.join(root, "linesegmentdetector-slam_amd", "liblsdhip_wdtest.so")
    if not os.path.exists(wd):
        subprocess.run(["make", "-C", os.path.join(root, "linesegmentdetector-slam_amd", "csrc"), "-j4", "wdtest"], check=True, capture_output=True)
    code = r"""
import ctypes as C, importlib, json, sys, numpy as np, torch
sys.path.insert(0, %r)
lsd = importlib.import_module("linesegmentdetector-slam_amd")
maps = np.load(%r)
img = np.ascontiguousarray(maps["aisle1"][:600, :800])
batch = np.stack([img] * 5)
c = lsd.Context(0)
out = {}
for waves in (4, 8):
    c.set_region_waves(waves)
    d = torch.from_numpy(batch).cuda()
    L = torch.zeros((5, 256, 10), dtype=torch.int64, device="cuda"); cnt = torch.zeros(5, dtype=torch.int32, device="cuda")
    ims = torch.zeros(batch.shape, dtype=torch.uint8, device="cuda")
    c.enqueue_device(d.data_ptr(), 5, 800, 600, L.data_ptr(), 256, cnt.data_ptr(), d_line_ims=ims.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    res = dict(counts=cnt.cpu().tolist(), lit=[int((ims[i] == 255).sum()) for i in range(5)])
    try:
        c.run_batch(batch.copy())
        res["status"] = 0
    except lsd.LsdError as e:
        res["status"] = e.status; res["msg"] = str(e)
    # the raw call: offsets and lines skip the images that were given up
    lp = C.c_void_p(); offs = (C.c_int * 6)(); p = lsd.make_params()
    b2 = batch.copy()
    res["st2"] = c.L.lsd_run_batch(c.h, b2.ctypes.data, 5, 800, 600, C.byref(p), None, C.byref(lp), offs)
    res["offs"] = list(offs); c.L.lsd_free(lp)
    out[waves] = res
print(json.dumps(out))
""" % (root, os.path.join(root, "tests", "golden", "maps.npz"))
    env = dict(os.environ, LSD_HIP_LIB=wd)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    import json
    got = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    img = np.ascontiguousarray(maps["aisle1"][:600, :800])
    lines, im = ctx.run(img.copy())
    n, lit = len(lines), int((im == 255).sum())
    assert n > 3
    for waves in ("4", "8"):
        r = got[waves]
        assert r["counts"] == [n, -1, n, -1, n], r
        assert r["lit"] == [lit, 0, lit, 0, lit], r
        assert r["status"] == lsdmod.LSD_ERR_INTERNAL and "image 1" in r["msg"], r
        assert r["st2"] == lsdmod.LSD_ERR_INTERNAL and r["offs"] == [0, n, n, 2 * n, 2 * n, 3 * n], r


def test_out_of_memory_is_reported_not_fatal(lsdmod, ctx):
    """A batch whose workspace cannot fit in HBM: LSD_ERR_NOMEM, and the context keeps working."""
    with pytest.raises(lsdmod.LsdError) as e:
        ctx.reserve(200000, 2048, 2048)                 # ~ 12 TB of workspace
    assert e.value.status == lsdmod.LSD_ERR_NOMEM
    lines, _ = ctx.run(np.zeros((64, 80), np.uint8))
    assert len(lines) == 0
