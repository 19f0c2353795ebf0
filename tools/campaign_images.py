"""The random images of the parity campaigns (tools/campaign.py, tests/golden/make_libm_ties.py): image i is a pure function of i."""
import numpy as np


def synth(rng, big=False):
    rows, cols = (int(rng.integers(1500, 3500)), int(rng.integers(1500, 3500))) if big else (int(rng.integers(60, 900)), int(rng.integers(60, 1200)))
    m = np.zeros((rows, cols), np.uint8)
    m[rng.random((rows, cols)) < rng.uniform(0.0, 0.5)] = 255
    for _ in range(int(rng.integers(3, 160 if big else 40))):
        x0, y0 = rng.integers(2, cols - 2), rng.integers(2, rows - 2)
        L = int(rng.integers(10, 2500 if big else 400)); a = rng.choice([0, np.pi / 2, np.pi / 4, rng.uniform(0, np.pi)])
        t = np.arange(L)
        xs = np.clip((x0 + t * np.cos(a)).astype(int), 0, cols - 1); ys = np.clip((y0 + t * np.sin(a)).astype(int), 0, rows - 1)
        m[ys, xs] = 1
        if rng.random() < 0.3:                                        # thick wall
            m[np.clip(ys + 1, 0, rows - 1), xs] = 1
    if rng.random() < 0.3:                                            # salt noise of occupied cells
        m[rng.random((rows, cols)) < 0.01] = 1
    return m


def campaign_image(i, big=False):
    """-> (image, parameters or {}, region-stage waves 0 / 4 / 8) of campaign image i"""
    rng = np.random.default_rng(10_000 + i)
    img = synth(rng, big)
    kw = {}
    if rng.random() < 0.3:
        kw = dict(sca=0.3, sig=float(rng.choice([0.6, 0.8])), angThre=float(rng.choice([22.5, 20.0, 30.0])),
                  denThre=float(rng.choice([0.7, 0.6])), pseBin=int(rng.choice([1024, 512, 256])))
    waves = int(rng.choice([0, 4, 8]))
    return img, kw, waves
