"""The C oracle (CPU-baseline / large-size checker) must agree with the NumPy
oracle that is pinned to the reference's goldens."""
import numpy as np
import pytest

from oracle import c_oracle as corc
from oracle import xrs_oracle as orc


def _dem(shape, seed=7, nan_frac=0.01):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:shape[0], 0:shape[1]]
    z = 2000 + 800 * np.sin(x / 30.0) * np.cos(y / 40.0) + rng.normal(0, 0.05, shape)
    z = z.astype(np.float32)
    if nan_frac:
        z[rng.random(shape) < nan_frac] = np.nan
    return z


@pytest.mark.parametrize("shape", [(37, 53), (3, 3), (2, 5), (128, 96)])
def test_terrain(shape):
    z = _dem(shape)
    np.testing.assert_array_equal(corc.slope(z, 30.0, 30.0), orc.slope(z, 30.0, 30.0))
    np.testing.assert_array_equal(corc.aspect(z), orc.aspect(z))
    np.testing.assert_array_equal(corc.curvature(z, 30.0), orc.curvature(z, 30.0))
    np.testing.assert_allclose(corc.hillshade(z), orc.hillshade(z), rtol=0, atol=5e-7, equal_nan=True)


def test_multispectral():
    rng = np.random.default_rng(3)
    a = rng.uniform(0, 3000, (64, 64)).astype(np.float32)
    b = rng.uniform(0, 3000, (64, 64)).astype(np.float32)
    c = rng.uniform(0, 3000, (64, 64)).astype(np.float32)
    a[0, 0] = b[0, 0] = 0
    a[1, 1] = np.nan
    np.testing.assert_array_equal(corc.normalized_ratio(a, b), orc.normalized_ratio(a, b))
    np.testing.assert_array_equal(corc.evi(a, b, c), orc.evi(a, b, c))
    np.testing.assert_array_equal(corc.savi(a, b, 0.5), orc.savi(a, b, 0.5))


@pytest.mark.parametrize("kernel", [orc.circle_kernel(1, 1, 2), orc.annulus_kernel(1, 1, 3, 1),
                                    np.array([[1, 0, 0], [0, 1, 0], [0, 0, 0]], dtype=float),
                                    np.array([[0.5, 1, 2.0]], dtype=float)])
def test_kxk(kernel):
    z = _dem((41, 35), nan_frac=0.02)
    # float64 accumulation order differs (tap-outer vs cell-outer) only in which adds happen
    # first per cell: both are row-major over taps, so results are bit-identical.
    np.testing.assert_array_equal(corc.convolve_2d(z, kernel), orc.convolve_2d(z, kernel))
    for stat in orc.FOCAL_STATS:
        np.testing.assert_array_equal(corc.focal_apply(z, kernel, stat), orc.focal_apply(z, kernel, stat),
                                      err_msg=stat)


def test_focal_mean3x3():
    z = _dem((33, 29), nan_frac=0.05).astype(np.float64)
    np.testing.assert_array_equal(corc.focal_mean3x3(z), orc.focal_mean3x3(z))
    np.testing.assert_array_equal(corc.focal_mean3x3(z, excludes=(np.nan, float(z[3, 3])), passes=3),
                                  orc.focal_mean3x3(z, excludes=(np.nan, float(z[3, 3])), passes=3))


def test_threads_do_not_change_results():
    z = _dem((64, 64))
    np.testing.assert_array_equal(corc.slope(z, 1, 1, nthreads=4), corc.slope(z, 1, 1, nthreads=1))
