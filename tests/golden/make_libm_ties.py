#!/usr/bin/env python3
"""Writes tests/golden/libm_ties.npz: the images of the random parity campaigns (tools/campaign.py, 160 000 images + 3 900 large ones,
profiles/r04u_campaign_*.log, profiles/r05h_campaign_*.log, profiles/r06r_campaign_*.log, profiles/r06t_campaign_*.log) on which the HIP path and the glibc-built oracle disagree by one accept / reject decision.  On every
one of them glibc misrounds a sin / cos / atan2 by one ulp on a structural tie (a rectangle edge exactly on a pixel row); the HIP
path evaluates those functions correctly rounded and equals the restatement rebuilt on correctly rounded functions
(oracle/liblsd_oracle_cr.so) bit for bit.  The images are pure functions of their campaign number (tools/campaign_images.py).
Also writes libm_ties.json: per image the parameters, the line counts of the two builds and in how many usedMap / lineIm pixels they differ
(what tests/test_oracle.py holds the two oracle builds to)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from campaign_images import campaign_image  # noqa: E402
from oracle import oracle  # noqa: E402

SMALL = [2995, 1759, 6382, 10647, 18262, 27264, 38359, 40212, 45813, 48150, 54902, 55115, 56831, 58631,
         68894, 73163, 75918, 79924, 80225, 80295, 83780,      # (the second row: images 60 000 .. 89 999, profiles/r05h_campaign_fresh30000.log)
         90756, 92641, 96504,                                   # (images 90 000 .. 109 999, round 6: profiles/r06r_campaign_fresh20000.log)
         117854, 118240, 130822, 135897, 136180, 143888, 144287, 146150, 147783]      # (images 100 000 .. 149 999: profiles/r06t_campaign_fresh50000.log)
BIG = [297, 2559]                                               # (2559: large images 2 000 .. 3 499, profiles/r06t_campaign_big_fresh1500.log)
# ... and the images of the campaigns with an NFA comparison inside, or within a factor two of, what an ulp of exp / log10 / pow can move (705: margin 0.27: two hopeless
# rectangles, 5 aligned pixels of 656, whose tails are 1 - 1e-15, so that logNFA = -logNT to the last place): both builds decide alike on
# it -- kept so that a libm (or a change of the device routines) that decides otherwise shows up
NEAR = [705, 1854, 2331, 3394]                            # (1854, 2331: margins 1.3 and 1.5, profiles/r05h_campaign_big_fresh600.log; 3394: 0.79, profiles/r06t_campaign_big_fresh1500.log)
NEAR_SMALL = [98908]                                      # (a small campaign image: two NFA values with a margin of 0.99, profiles/r06r_campaign_fresh20000.log)
NAMES = {2995: "tie_a", 1759: "tie_b"}                     # (the two fixtures of round 3 keep their names)

out, table = {}, {}
for i, big in [(i, False) for i in SMALL] + [(i, True) for i in BIG] + [(i, True) for i in NEAR] + [(i, False) for i in NEAR_SMALL]:
    img, kw, _ = campaign_image(i, big)
    name = NAMES.get(i, "%s%d" % ("near" if (i in NEAR and big) or (i in NEAR_SMALL and not big) else "big" if big else "img", i))
    a = oracle.lsd(img.copy(), debug=True, **kw)
    b = oracle.lsd(img.copy(), debug=True, _lib=oracle.lib_cr(), **kw)
    out[name] = img
    table[name] = (kw, len(a["lines"]), len(b["lines"]), int((a["dbg"]["used"] != b["dbg"]["used"]).sum()),
                   int((a["lineIm"] != b["lineIm"]).sum()))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "libm_ties.npz"), **out)
json.dump({k: dict(params=v[0], lines_glibc=v[1], lines_cr=v[2], used_diff=v[3], lineim_diff=v[4]) for k, v in table.items()},
          open(os.path.join(ROOT, "tests", "golden", "libm_ties.json"), "w"), indent=1)
for k, v in table.items():
    print(k, v)
