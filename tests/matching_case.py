"""A synthetic scan-to-map matching frame built from a fixture map (shared by the CPU and GPU tests)."""
import numpy as np


def build_case(map_u8, res, oracle, theta_deg=17.0, centre=(300.0, 260.0), half=140, max_points=1500, seed=3):
    """Map lines + mapCache from the oracle; the 'scan' is the window of the map around `centre`, expressed in a scan
    image frame that is rotated by -theta and shifted, so that the true candidate pose has angDiff = theta."""
    rng = np.random.default_rng(seed)
    map_cache = oracle.map_cache(map_u8.copy(), res)
    map_lines = oracle.lsd(map_u8.copy())["lines"]
    th = np.deg2rad(theta_deg)
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    origin = np.array(centre) - half                          # map point that becomes (0, 0) of the scan frame before rotation
    to_scan = lambda p: (np.asarray(p, float) - origin - half) @ R + half     # rotate by -theta about the window centre
    ys, xs = np.nonzero(map_u8 == 1)
    sel = (np.abs(xs - centre[0]) < half) & (np.abs(ys - centre[1]) < half)
    pts_map = np.stack([xs[sel], ys[sel]], 1).astype(float)
    if len(pts_map) > max_points:
        pts_map = pts_map[rng.choice(len(pts_map), max_points, replace=False)]
    pts = np.zeros((len(pts_map), 3))
    pts[:, :2] = to_scan(pts_map)
    inside = lambda l: all(abs(l[a] - centre[0]) < half and abs(l[b] - centre[1]) < half for a, b in (("x1", "y1"), ("x2", "y2")))
    scan_lines = np.array([l for l in map_lines if inside(l)], dtype=map_lines.dtype)
    for l in scan_lines:
        (l["x1"], l["y1"]), (l["x2"], l["y2"]) = to_scan((l["x1"], l["y1"])), to_scan((l["x2"], l["y2"]))
        l["dx"], l["dy"] = l["x2"] - l["x1"], l["y2"] - l["y1"]
    lidar_scan = np.array([half, half], float)                # the lidar sits at the window centre
    return dict(map_cache=map_cache, map_lines=map_lines, scan_lines=scan_lines, pts=pts, lidar=(half, half, 0.0),
                lidar_map=np.array(centre, float), theta=theta_deg)
