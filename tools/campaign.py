#!/usr/bin/env python3
"""Developer tool: randomized parity campaign -- many synthetic maps (walls, noise, random sizes, random parameters) through the
HIP path and the oracle; reports every image whose usedMap, line count, lineIm or line records differ."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: F401  (before the HIP library)
from oracle import oracle
lsd = importlib.import_module("linesegmentdetector-slam_amd")
oracle.build()
ctx = lsd.Context(0)
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 200
only = [int(x) for x in sys.argv[2:]]                      # optional: just these image numbers


from campaign_images import campaign_image                  # image i is a pure function of i (shared with tests/golden/make_libm_ties.py)
BIG = bool(os.environ.get("CAMPAIGN_BIG"))                    # larger maps (spill paths, long lists): CAMPAIGN_BIG=1
# The floor the campaign ENFORCES on RectangleImprover's comparisons (logNFA > 0, candidate > best so far).  The HIP path evaluates
# the NFA's exp / log10 / pow correctly rounded; glibc's are within one ulp of that (its log10 differs in one call out of seven:
# tests/test_crmath.py).  The region stage records every comparison's MARGIN: the distance of its operands over the most those last
# places can move them apart (k_region.hip: improve()).  Below 1 a decision could differ between the two libms; the campaign fails
# below kMarginFloor.  (What comes closest are structural near-ties: B(1/p + 1, 1/p, p) = p^(1/p - 1) exactly, evaluated once through
# the log-gamma formulas and once in closed form, ~3e-13 apart = a margin of ~25; and tails of almost 1, where logNFA = -logNT + 1e-14.)
kMarginFloor = 2.0

CR_EVERY = int(os.environ.get("CAMPAIGN_CR", "10"))          # every n-th image also against the correctly rounded restatement, NFA values to the bit
bad = cr_bad = cr_n = 0
nfa_abs, nfa_gap = float('inf'), float('inf')      # smallest margins of the campaign's NFA comparisons
t0 = time.time()
first = int(os.environ.get("CAMPAIGN_FIRST", "0"))           # image numbers first .. first + n - 1 (fresh images: past what earlier campaigns saw)
for i in (only or range(first, first + n_img)):
    img, kw, waves = campaign_image(i, BIG)
    ctx.set_region_waves(waves)
    ref = oracle.lsd(img.copy(), debug=True, **kw)
    d = ref["dbg"]
    with_cr = CR_EVERY > 0 and i % CR_EVERY == 0
    ctx.set_trace(with_cr)
    lines, im = ctx.run(img.copy(), lsd.make_params(**kw) if kw else None)
    used = (ctx.fetch(0, lsd.DBG_STATE, (d["w"], d["h"])) & 3).astype(np.uint8)
    st_ = ctx.fetch(0, lsd.DBG_STATS, (d["w"], d["h"]))
    if with_cr:                                               # the seed trace against the restatement on correctly rounded functions
        rcs = oracle.lsd(img.copy(), debug=True, _lib=oracle.lib_cr(), **kw)["dbg"]["seeds"]
        sd = ctx.fetch(0, lsd.DBG_SEEDS, (d["w"], d["h"]))
        cr_n += 1
        if not (len(sd) == len(rcs) and all(np.array_equal(sd[f], rcs[f]) for f in ("order_idx", "num", "outcome", "final_num", "logNFA"))):
            cr_bad += 1
            print("CR-MISMATCH image", i, img.shape, kw, "seed trace differs from the correctly rounded restatement", flush=True)
            np.save(os.path.join(ROOT, "gpurun_out", "campaign_crbad_%d.npy" % i), img)
    nfa_abs, nfa_gap = min(nfa_abs, st_["nfa_min_abs"]), min(nfa_gap, st_["nfa_min_gap"])
    ok = len(lines) == len(ref["lines"]) and np.array_equal(used, d["used"]) and np.array_equal(im, ref["lineIm"])
    if ok and len(lines):
        ok = all(np.abs(lines[f] - ref["lines"][f]).max() < 1e-6 for f in ("x1", "y1", "x2", "y2")) and np.array_equal(lines["orient"], ref["lines"]["orient"])
    if not ok:
        bad += 1
        # the diagnostic variant with correctly rounded sin/cos/atan2 (oracle/cr_shim.cpp): a libm tie if the HIP path equals it
        rc = oracle.lsd(img.copy(), debug=True, _lib=oracle.lib_cr(), **kw)
        eq = (len(lines) == len(rc["lines"]) and np.array_equal(used, rc["dbg"]["used"]) and np.array_equal(im, rc["lineIm"]) and
              (len(lines) == 0 or all(np.abs(lines[f] - rc["lines"][f]).max() < 1e-6 for f in ("x1", "y1", "x2", "y2"))))
        print("MISMATCH image", i, img.shape, kw, "lines", len(lines), "vs", len(ref["lines"]), "usedMap diff", int((used != d["used"]).sum()),
              "| equals the correctly rounded restatement:", eq, flush=True)
        if not eq:
            np.save(os.path.join(ROOT, "gpurun_out", "campaign_bad_%d.npy" % i), img)
print("campaign: %d images, %d mismatches, %.0f s; smallest margins of the NFA comparisons (distance of the operands / what one ulp of exp, log10, pow can "
      "move them): logNFA against 0: %.3g, two NFA values: %.3g; enforced floor %.1f (values made of host constants alone -- -logNT - n log10 p, "
      "which is exactly 0 for w h = 6^4, p = 1/6, n = 10 -- are the reference's own numbers and not counted against 0)" % (n_img, bad, time.time() - t0, nfa_abs, nfa_gap, kMarginFloor))
print("          %d of them also against the restatement on correctly rounded functions, every seed's decision and logNFA bit for bit: %d differ" % (cr_n, cr_bad))
if cr_bad:
    sys.exit(1)
if min(nfa_abs, nfa_gap) < kMarginFloor:
    print("FAIL: an NFA comparison's margin is below the floor: a decision could differ between correctly rounded functions and glibc's")
    sys.exit(1)
