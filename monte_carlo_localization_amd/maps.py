"""Occupancy-grid inputs for the engine's harness (bench / tests / demo).

The reference never reads map files itself: nav2 `map_server` turns `<name>.yaml` + image into a
`nav_msgs/OccupancyGrid` and the node fetches it with a GetMap call (cpp:184-190,
launch/mcl_launch.py:62-71).  nav2_map_server is not part of the reference tree and its version
is not pinned there, so this loader follows the documented default (`mode: trinary`) rule as
recorded in SURVEY.md Appendix C — "parity unpinned" for this step:

    shade = mean of colour channels (alpha ignored), 0..255
    occ   = shade/255 if negate else (255-shade)/255
    occ > occupied_thresh -> 100 ; occ < free_thresh -> 0 ; else -1
    image row 0 is the TOP of the map, grid row 0 the BOTTOM (vertical flip); data[row*W+col]

The other two documented modes are provided for maps saved with them (same status: restated from the
nav2 documentation, not pinned by anything in the reference):
    scale:  transparent pixel -> -1; occ > occupied_thresh -> 100; occ < free_thresh -> 0;
            else 99 * (occ - free_thresh) / (occupied_thresh - free_thresh), rounded to nearest
    raw:    value = round(occ * 255) when that is in 0..100, else -1
The hot path only distinguishes data > 50 (cpp:642) and data == 0 (free list, cpp:199-213).
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np


@dataclass
class OccupancyMap:
    """The subset of nav_msgs/OccupancyGrid the hot path reads (cpp:190-195, 628-642)."""
    data: np.ndarray          # int8, shape (H, W), row 0 = bottom; values {-1, 0, 100}
    resolution: np.float32    # MapMetaData.resolution is float32
    origin_x: float
    origin_y: float
    name: str = ""

    @property
    def width(self) -> int:
        return int(self.data.shape[1])

    @property
    def height(self) -> int:
        return int(self.data.shape[0])


def trinary_from_image(img: np.ndarray, negate: bool, occupied_thresh: float, free_thresh: float) -> np.ndarray:
    a = np.asarray(img)
    if a.ndim == 3:
        ch = a.shape[2]
        if ch in (2, 4):            # drop alpha
            a = a[..., : ch - 1]
        shade = a.astype(np.float64).mean(axis=2)
    else:
        shade = a.astype(np.float64)
    occ = shade / 255.0 if negate else (255.0 - shade) / 255.0
    grid = np.full(occ.shape, -1, dtype=np.int8)
    grid[occ > occupied_thresh] = 100
    grid[occ < free_thresh] = 0
    return np.ascontiguousarray(grid[::-1])


def grid_from_image(img: np.ndarray, mode: str = "trinary", negate: bool = False, occupied_thresh: float = 0.65,
                    free_thresh: float = 0.196) -> np.ndarray:
    """Image (grey, grey+alpha, RGB or RGBA; 8 or 16 bit) -> int8 occupancy grid, bottom row first."""
    if mode == "trinary":
        return trinary_from_image(img, negate, occupied_thresh, free_thresh)
    a = np.asarray(img)
    full = float(np.iinfo(a.dtype).max) if np.issubdtype(a.dtype, np.integer) else 1.0
    alpha = None
    if a.ndim == 3:
        ch = a.shape[2]
        if ch in (2, 4):
            alpha = a[..., ch - 1].astype(np.float64) / full
            a = a[..., : ch - 1]
        shade = a.astype(np.float64).mean(axis=2) / full
    else:
        shade = a.astype(np.float64) / full
    occ = shade if negate else 1.0 - shade
    if mode == "scale":
        ratio = (occ - free_thresh) / (occupied_thresh - free_thresh)
        grid = np.rint(99.0 * ratio).clip(1, 99).astype(np.int8)
        grid[occ > occupied_thresh] = 100
        grid[occ < free_thresh] = 0
        if alpha is not None:
            grid[alpha < 1.0] = -1
    elif mode == "raw":
        v = np.rint(occ * 255.0)
        grid = np.where((v >= 0) & (v <= 100), v, -1).astype(np.int8)
    else:
        raise ValueError(f"unknown map mode {mode!r} (trinary, scale, raw)")
    return np.ascontiguousarray(grid[::-1])


def load_map_yaml(yaml_path: str) -> OccupancyMap:
    """<name>.yaml + PNG/PGM/BMP image as nav2 map_server reads them (image, mode, resolution, origin, negate,
    occupied_thresh, free_thresh)."""
    import yaml
    from PIL import Image
    with open(yaml_path) as f:
        meta = yaml.safe_load(f)
    img_path = meta["image"] if os.path.isabs(meta["image"]) else os.path.join(os.path.dirname(yaml_path), meta["image"])
    img = np.array(Image.open(img_path))
    grid = grid_from_image(img, str(meta.get("mode", "trinary")), bool(int(meta.get("negate", 0))),
                           float(meta.get("occupied_thresh", 0.65)), float(meta.get("free_thresh", 0.196)))
    org = meta["origin"]
    return OccupancyMap(grid, np.float32(meta["resolution"]), float(org[0]), float(org[1]),
                        os.path.splitext(os.path.basename(yaml_path))[0])


def save_npz(m: OccupancyMap, path: str) -> None:
    """Compact fixture: occupied / unknown masks bit-packed."""
    np.savez_compressed(path, occ=np.packbits(m.data > 50), unk=np.packbits(m.data < 0),
                        shape=np.array(m.data.shape, np.int32), resolution=np.float32(m.resolution),
                        origin=np.array([m.origin_x, m.origin_y], np.float64), name=np.array(m.name))


def load_npz(path: str) -> OccupancyMap:
    z = np.load(path)
    H, W = (int(v) for v in z["shape"])
    occ = np.unpackbits(z["occ"])[: H * W].reshape(H, W).astype(bool)
    unk = np.unpackbits(z["unk"])[: H * W].reshape(H, W).astype(bool)
    grid = np.zeros((H, W), np.int8)
    grid[occ] = 100
    grid[unk] = -1
    return OccupancyMap(grid, np.float32(z["resolution"]), float(z["origin"][0]), float(z["origin"][1]),
                        str(z["name"]))


def synthetic_levine(width: int = 2049, height: int = 2049, seed: int = 7) -> OccupancyMap:
    """Stand-in for maps/levine.pgm, which is absent from the reference tree
    (.MISSING_LARGE_BLOBS:1; only levine.yaml exists: resolution 0.05, origin -51.224998).
    SYNTHETIC: a building-like loop of 2.5 m wide corridors with side rooms and door gaps,
    unknown space outside the walls."""
    rng = np.random.default_rng(seed)
    g = np.full((height, width), -1, np.int8)

    def free(y0, y1, x0, x1):
        g[y0:y1, x0:x1] = 0

    cw = 50  # corridor width in cells (2.5 m)
    m0, m1 = 300, width - 300
    # outer corridor loop + two cross corridors
    free(m0, m0 + cw, m0, m1); free(m1 - cw, m1, m0, m1)
    free(m0, m1, m0, m0 + cw); free(m0, m1, m1 - cw, m1)
    mid = width // 2
    free(m0, m1, mid - cw // 2, mid + cw // 2)
    free(mid - cw // 2, mid + cw // 2, m0, m1)
    # rooms hanging off the corridors
    for _ in range(60):
        w, h = rng.integers(60, 160, size=2)
        x = int(rng.integers(m0 - 150, m1 + 150 - w)); y = int(rng.integers(m0 - 150, m1 + 150 - h))
        free(y, y + h, x, x + w)
    # walls: every unknown cell 4-adjacent to free space becomes occupied (1-cell shell), thickened once
    for _ in range(2):
        f = g == 0 if _ == 0 else g == 100
        nb = np.zeros_like(f)
        nb[1:, :] |= f[:-1, :]; nb[:-1, :] |= f[1:, :]; nb[:, 1:] |= f[:, :-1]; nb[:, :-1] |= f[:, 1:]
        g[(g == -1) & nb] = 100
    # clutter inside free space
    fy, fx = np.nonzero(g == 0)
    pick = rng.choice(fy.size, size=400, replace=False)
    for k in pick:
        y, x = int(fy[k]), int(fx[k])
        g[y:y + 3, x:x + 3] = 100
    # keep the start pose clear: corridor centre near the bottom-left
    return OccupancyMap(np.ascontiguousarray(g), np.float32(0.05), -51.224998, -51.224998, "levine_synthetic")


def synthetic_fine025(base: OccupancyMap) -> OccupancyMap:
    """SYNTHETIC long-range case: `base`'s grid with every cell split into 2 x 2, declared at 0.025 m per cell.  At the stock
    12 m range that is MAX_RANGE_PX = 479 (cpp:195 puts no bound on it; the float32 resolution is a hair above 0.025, SURVEY
    D9) -- twice what the reference's own maps give.  The origin keeps pixel (2 * 1464 + 1, 2 * 626 + 1) -- the start / finish
    straight of Spielberg_map -- at world (0, 0)."""
    g = np.ascontiguousarray(np.repeat(np.repeat(base.data, 2, axis=0), 2, axis=1))
    res = np.float32(0.025)
    return OccupancyMap(g, res, -(2 * 1464 + 1) * float(res), -(2 * 626 + 1) * float(res), base.name + "_x2_at_0.025")
