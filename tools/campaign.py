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

CR_EVERY = int(os.environ.get("CAMPAIGN_CR", "10"))          # every n-th image also SEED BY SEED against the correctly rounded restatement, NFA values to the bit
CR_ALL = os.environ.get("CAMPAIGN_CR_ALL", "1") != "0"        # every image's usedMap / lineIm / lines also against the correctly rounded restatement
# Large maps on which an NFA comparison's margin is below the floor and BOTH oracle builds decide alike (fixtures near705, near1854,
# near2331, near3394, near98908 of tests/golden/libm_ties.npz): known, listed, and the only ones allowed -- a NEW image below the floor fails the campaign.
ALLOW_MARGIN = {(True, 705), (True, 1854), (True, 2331), (True, 3394), (False, 98908)}      # (98908: found by the fresh images of round 6 -- the check doing its work -- and a fixture since)
bad = cr_bad = cr_n = hard = 0
prop_bad = sens = margin_bad = 0
nfa_abs, nfa_gap = float('inf'), float('inf')      # smallest margins of the campaign's NFA comparisons
t0 = time.time()
first = int(os.environ.get("CAMPAIGN_FIRST", "0"))           # image numbers first .. first + n - 1 (fresh images: past what earlier campaigns saw)


def same(lines, used, im, r):
    return (len(lines) == len(r["lines"]) and np.array_equal(used, r["dbg"]["used"]) and np.array_equal(im, r["lineIm"]) and
            (len(lines) == 0 or (all(np.abs(lines[f] - r["lines"][f]).max() < 1e-6 for f in ("x1", "y1", "x2", "y2")) and
                                 np.array_equal(lines["orient"], r["lines"]["orient"]))))


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
    ties = int(ctx.last_sensitivity(1)[0])                    # decisions within the libm's noise (lsd_last_sensitivity)
    sens += ties > 0
    rc = oracle.lsd(img.copy(), debug=True, _lib=oracle.lib_cr(), **kw) if (CR_ALL or with_cr) else None
    if with_cr:                                               # the seed trace against the restatement on correctly rounded functions
        rcs = rc["dbg"]["seeds"]
        sd = ctx.fetch(0, lsd.DBG_SEEDS, (d["w"], d["h"]))
        cr_n += 1
        if not (len(sd) == len(rcs) and all(np.array_equal(sd[f], rcs[f]) for f in ("order_idx", "num", "outcome", "final_num", "logNFA"))):
            cr_bad += 1
            print("CR-MISMATCH image", i, img.shape, kw, "seed trace differs from the correctly rounded restatement", flush=True)
            np.save(os.path.join(ROOT, "gpurun_out", "campaign_crbad_%d.npy" % i), img)
    if rc is not None and not same(lines, used, im, rc):      # the HIP path IS the correctly rounded restatement, on every image
        hard += 1
        print("HARD-MISMATCH image", i, img.shape, kw, "differs from the correctly rounded restatement: lines", len(lines), "vs", len(rc["lines"]),
              "usedMap diff", int((used != rc["dbg"]["used"]).sum()), flush=True)
        np.save(os.path.join(ROOT, "gpurun_out", "campaign_bad_%d.npy" % i), img)
    m_abs, m_gap = st_["nfa_min_abs"], st_["nfa_min_gap"]
    nfa_abs, nfa_gap = min(nfa_abs, m_abs), min(nfa_gap, m_gap)
    if min(m_abs, m_gap) < kMarginFloor:
        listed = (BIG, i) in ALLOW_MARGIN
        print("MARGIN image", i, img.shape, "smallest margins %.3g / %.3g, near ties %d:" % (m_abs, m_gap, ties), "on the allow-list" if listed else "NOT on the allow-list", flush=True)
        margin_bad += 0 if listed else 1
        if ties == 0:
            prop_bad += 1
            print("PROPERTY image", i, "a margin below the floor with a near-tie count of 0", flush=True)
    if not same(lines, used, im, ref):
        bad += 1
        if rc is None:
            rc = oracle.lsd(img.copy(), debug=True, _lib=oracle.lib_cr(), **kw)
        eq = same(lines, used, im, rc)
        print("MISMATCH image", i, img.shape, kw, "lines", len(lines), "vs", len(ref["lines"]), "usedMap diff", int((used != d["used"]).sum()),
              "| equals the correctly rounded restatement:", eq, "| near ties:", ties, flush=True)
        if not eq and not (CR_ALL or with_cr):                   # (otherwise counted above already)
            hard += 1
            np.save(os.path.join(ROOT, "gpurun_out", "campaign_bad_%d.npy" % i), img)
        if ties == 0:                                         # the property: glibc and the correctly rounded functions can only part at a counted decision
            prop_bad += 1
            print("PROPERTY image", i, "differs from the glibc build with a near-tie count of 0", flush=True)
print("campaign: %d images, %d differ from the glibc build (libm ties: each equals the correctly rounded restatement and has a near-tie count > 0), %d differ from the "
      "correctly rounded restatement, %.0f s; %d images (%.2f %%) have a near-tie count > 0 (lsd_last_sensitivity), every other one equals the glibc build; "
      "smallest margins of the NFA comparisons (distance of the operands / what one ulp of exp, log10, pow can "
      "move them): logNFA against 0: %.3g, two NFA values: %.3g; enforced floor %.1f off the allow-list (values made of host constants alone -- -logNT - n log10 p, "
      "which is exactly 0 for w h = 6^4, p = 1/6, n = 10 -- are the reference's own numbers and not counted against 0)" % (
          n_img, bad, hard, time.time() - t0, sens, 100.0 * sens / max(1, n_img), nfa_abs, nfa_gap, kMarginFloor))
print("          %d of them also against the restatement on correctly rounded functions, every seed's decision and logNFA bit for bit: %d differ" % (cr_n, cr_bad))
fail = []
if hard:
    fail.append("%d image(s) differ from the correctly rounded restatement" % hard)
if cr_bad:
    fail.append("%d seed trace(s) differ from the correctly rounded restatement" % cr_bad)
if prop_bad:
    fail.append("%d image(s) differ from the glibc build (or sit below the margin floor) with a near-tie count of 0" % prop_bad)
if margin_bad:
    fail.append("%d image(s) off the allow-list have an NFA comparison below the margin floor" % margin_bad)
print("FAIL: " + "; ".join(fail) if fail else "PASS: every image equals the correctly rounded restatement; every image that differs from the glibc build has a near-tie count > 0; "
      "no image off the allow-list is below the margin floor")
sys.exit(1 if fail else 0)
