"""Deterministic synthetic rasters shared by tests and bench (SURVEY.md §8d recipes)."""
import numpy as np


def smooth_dem(shape, seed=7, nan_frac=0.0, cellsize=30.0):
    """Parity-stress DEM: large smooth values + small noise exposes float32 cancellation."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:shape[0], 0:shape[1]]
    z = 2000 + 800 * np.sin(x / 300.0 * cellsize / 3) * np.cos(y / 400.0 * cellsize / 3)
    z = (z + rng.normal(0, 0.05, shape)).astype(np.float32)
    if nan_frac:
        z[rng.random(shape) < nan_frac] = np.nan
    return z


def asv_dem(rows, cols, seed=71942, y0=0, total_rows=None, nan_frac=0.0):
    """The reference's own benchmark raster (benchmarks/benchmarks/common.py:26-35) at any size:
    100*exp(-x^2/5e5 - y^2/2e5) + N(0, 2); rows [y0, y0+rows) of a `total_rows`-row raster.
    nan_frac: that share of the cells, scattered (seeded by the band's first row), is nodata (0.1 %)."""
    total_rows = total_rows or rows
    x = np.linspace(-180, 180, cols)
    y = np.linspace(-90, 90, total_rows)[y0:y0 + rows]
    x2, y2 = np.meshgrid(x, y)
    rng = np.random.default_rng(seed + y0)
    z = (100.0 * np.exp(-x2 ** 2 / 5e5 - y2 ** 2 / 2e5) + rng.normal(0.0, 2.0, (rows, cols))).astype(np.float32)
    if nan_frac:
        scatter_nodata(z, nan_frac, seed + 7919 + y0)
    return z


def scatter_nodata(z, frac, seed):
    """NaN into round(frac * size) cells of z (in place), positions drawn with replacement from a seeded generator."""
    n = int(round(frac * z.size))
    if n:
        z.reshape(-1)[np.random.default_rng(seed).integers(0, z.size, n)] = np.nan
    return z


def bands(shape, seed):
    rng = np.random.default_rng(seed)
    return (500 + 100 * rng.random(shape) + rng.normal(0, 2.0, shape)).astype(np.float32)


def block_zones(rows, cols, n_zones=1000, block=1024, y0=0):
    i = (np.arange(y0, y0 + rows) // block)[:, None]
    j = (np.arange(cols) // block)[None, :]
    return ((i * 32 + j) % n_zones).astype(np.int32)
