/*
 * CPU oracle, C restatement (TEST INFRASTRUCTURE ONLY -- the checker, never the product).
 *
 * Scalar loops that follow the reference's Numba CPU kernels cell by cell, with
 * Numba's typing written out (int literal * float32 -> double; float32 op
 * float32 -> float; Python-float argument -> double).  Used (a) to cross-check
 * the NumPy restatement in oracle/xrs_oracle.py, which is the one pinned to the
 * reference's golden vectors, and (b) as the `cpu_baseline` ("port") leg of
 * bench.py, because it runs at Numba-like speed on large rasters where the
 * NumPy restatement would need many full-size temporaries.
 *
 * Build: oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).  No fast-math,
 * no FMA contraction: Numba does not contract either.
 *
 * `nthreads` = 1 reproduces the reference (its kernels are single-threaded:
 * `prange` without parallel=True, xrspatial/utils.py:31); >1 splits output rows
 * with OpenMP, which is what the reference's dask threaded scheduler amounts to.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#define IDX(y, x) ((size_t)(y) * (size_t)cols + (size_t)(x))

static void fill_nan_f32(float *out, size_t n) {
    for (size_t i = 0; i < n; ++i) out[i] = NAN;
}

/* xrspatial/slope.py:56-76 */
void orc_slope(const float *data, float *out, int rows, int cols,
               double cellsize_x, double cellsize_y, int nthreads) {
    fill_nan_f32(out, (size_t)rows * cols);
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int y = 1; y < rows - 1; ++y) {
        for (int x = 1; x < cols - 1; ++x) {
            float a = data[IDX(y + 1, x - 1)], b = data[IDX(y + 1, x)], c = data[IDX(y + 1, x + 1)];
            float d = data[IDX(y, x - 1)], f = data[IDX(y, x + 1)];
            float g = data[IDX(y - 1, x - 1)], h = data[IDX(y - 1, x)], i = data[IDX(y - 1, x + 1)];
            double dz_dx = (((double)c + 2.0 * (double)f + (double)i) -
                            ((double)a + 2.0 * (double)d + (double)g)) / (8 * cellsize_x);
            double dz_dy = (((double)g + 2.0 * (double)h + (double)i) -
                            ((double)a + 2.0 * (double)b + (double)c)) / (8 * cellsize_y);
            double p = pow(dz_dx * dz_dx + dz_dy * dz_dy, .5);
            out[IDX(y, x)] = (float)(atan(p) * 57.29578);
        }
    }
}

/* xrspatial/aspect.py:56-90 */
void orc_aspect(const float *data, float *out, int rows, int cols, int nthreads) {
    const double RADIAN = 180 / M_PI;
    fill_nan_f32(out, (size_t)rows * cols);
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int y = 1; y < rows - 1; ++y) {
        for (int x = 1; x < cols - 1; ++x) {
            float a = data[IDX(y - 1, x - 1)], b = data[IDX(y - 1, x)], c = data[IDX(y - 1, x + 1)];
            float d = data[IDX(y, x - 1)], f = data[IDX(y, x + 1)];
            float g = data[IDX(y + 1, x - 1)], h = data[IDX(y + 1, x)], i = data[IDX(y + 1, x + 1)];
            double dz_dx = (((double)c + 2.0 * (double)f + (double)i) -
                            ((double)a + 2.0 * (double)d + (double)g)) / 8;
            double dz_dy = (((double)g + 2.0 * (double)h + (double)i) -
                            ((double)a + 2.0 * (double)b + (double)c)) / 8;
            if (dz_dx == 0 && dz_dy == 0) {
                out[IDX(y, x)] = -1.f;
            } else {
                double asp = atan2(dz_dy, -dz_dx) * RADIAN;
                if (asp < 0) out[IDX(y, x)] = (float)(90.0 - asp);
                else if (asp > 90.0) out[IDX(y, x)] = (float)(360.0 - asp + 90.0);
                else out[IDX(y, x)] = (float)(90.0 - asp);
            }
        }
    }
}

/* xrspatial/curvature.py:31-41 */
void orc_curvature(const float *data, float *out, int rows, int cols, double cellsize, int nthreads) {
    fill_nan_f32(out, (size_t)rows * cols);
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int y = 1; y < rows - 1; ++y) {
        for (int x = 1; x < cols - 1; ++x) {
            float vs = data[IDX(y + 1, x)] + data[IDX(y - 1, x)];      /* f32 + f32 */
            float hs = data[IDX(y, x + 1)] + data[IDX(y, x - 1)];
            double d = (double)vs / 2 - (double)data[IDX(y, x)];
            double e = (double)hs / 2 - (double)data[IDX(y, x)];
            out[IDX(y, x)] = (float)(-2 * (d + e) * 100 / (cellsize * cellsize));
        }
    }
}

/* xrspatial/hillshade.py:20-35.  The reference is vectorised NumPy whose float32
 * sin/cos/arctan ufuncs are SIMD routines; libm's sinf/cosf/atanf may differ from
 * them by an ulp, so this port agrees with the NumPy restatement to ~1e-7 absolute,
 * not bit for bit.  Output float64 like the reference under NumPy >= 2. */
void orc_hillshade(const float *data, double *out, int rows, int cols,
                   double azimuth, double angle_altitude, int nthreads) {
    double az = 360.0 - azimuth;
    double azr = az * M_PI / 180.;
    double alr = angle_altitude * M_PI / 180.;
    double sin_alt = sin(alr), cos_alt = cos(alr);
    float half_pi = (float)(M_PI / 2.);
    float az_off = (float)(azr - M_PI / 2.);
    for (size_t i = 0; i < (size_t)rows * cols; ++i) out[i] = NAN;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int y = 1; y < rows - 1; ++y) {
        for (int x = 1; x < cols - 1; ++x) {
            float gy = (data[IDX(y + 1, x)] - data[IDX(y - 1, x)]) / 2.0f;
            float gx = (data[IDX(y, x + 1)] - data[IDX(y, x - 1)]) / 2.0f;
            float slope = half_pi - atanf(sqrtf(gy * gy + gx * gx));
            float aspect = atan2f(-gy, gx);
            double shaded = sin_alt * (double)sinf(slope) +
                            cos_alt * (double)cosf(slope) * (double)cosf(az_off - aspect);
            out[IDX(y, x)] = (shaded + 1) / 2;
        }
    }
}

/* xrspatial/multispectral.py:825-841 */
void orc_normalized_ratio(const float *a, const float *b, float *out, size_t n, int nthreads) {
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (size_t i = 0; i < n; ++i) {
        float num = a[i] - b[i], den = a[i] + b[i];
        out[i] = (den == 0.0f) ? NAN : num / den;
    }
}

/* xrspatial/multispectral.py:175-188 */
void orc_evi(const float *nir, const float *red, const float *blue, float *out, size_t n,
             double c1, double c2, double soil_factor, double gain, int nthreads) {
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (size_t i = 0; i < n; ++i) {
        float num = nir[i] - red[i];
        double den = (double)nir[i] + c1 * (double)red[i] - c2 * (double)blue[i] + soil_factor;
        out[i] = (den != 0.0) ? (float)(gain * ((double)num / den)) : NAN;
    }
}

/* xrspatial/multispectral.py:876-890 */
void orc_savi(const float *nir, const float *red, float *out, size_t n, double soil_factor, int nthreads) {
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (size_t i = 0; i < n; ++i) {
        float num = nir[i] - red[i];
        double soma = (double)(nir[i] + red[i]) + soil_factor;
        double den = soma * (1.0 + soil_factor);
        out[i] = (den != 0.0) ? (float)((double)num / den) : NAN;
    }
}

/* xrspatial/convolution.py:285-313 */
void orc_convolve2d(const float *data, float *out, int rows, int cols,
                    const double *kernel, int krows, int kcols, int nthreads) {
    int wkx = krows / 2, wky = kcols / 2;
    fill_nan_f32(out, (size_t)rows * cols);
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int i = wkx; i < rows - wkx; ++i) {
        for (int j = wky; j < cols - wky; ++j) {
            double num = 0.0;
            for (int ii = i - wkx; ii < i + wkx + 1; ++ii)
                for (int jj = j - wky; jj < j + wky + 1; ++jj)
                    num += kernel[(wkx + ii - i) * kcols + (wky + jj - j)] * (double)data[IDX(ii, jj)];
            out[IDX(i, j)] = (float)num;
        }
    }
}

/* xrspatial/focal.py:44-67 (float64 in/out, one pass) */
void orc_focal_mean3x3(const double *data, double *out, int rows, int cols,
                       const double *excludes, int nexcl, int nthreads) {
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int y = 0; y < rows; ++y) {
        for (int x = 0; x < cols; ++x) {
            double v = data[IDX(y, x)];
            int excl = 0;
            for (int e = 0; e < nexcl; ++e)
                if (v == excludes[e] || (isnan(v) && isnan(excludes[e]))) { excl = 1; break; }
            if (excl) { out[IDX(y, x)] = v; continue; }
            int l = x - 1 < 0 ? 0 : x - 1, r = x + 2 > cols ? cols : x + 2;
            int b = y - 1 < 0 ? 0 : y - 1, t = y + 2 > rows ? rows : y + 2;
            double c = 0.0; long cnt = 0;
            for (int yy = b; yy < t; ++yy)
                for (int xx = l; xx < r; ++xx) {
                    double w = data[IDX(yy, xx)];
                    if (!isnan(w)) { c += w; ++cnt; }
                }
            out[IDX(y, x)] = c / (double)cnt;     /* 0/0 -> NaN, like np.divide */
        }
    }
}

/* xrspatial/focal.py:305-326 with the built-in reducers focal.py:268-302.
 * stat: 0 mean, 1 max, 2 min, 3 range, 4 std, 5 var, 6 sum. */
static float reduce_window(const float *w, int n, int stat) {
    /* numba nan-reductions, numba/np/arraymath.py (see oracle/xrs_oracle.py docstring) */
    if (stat == 6) {
        float c = 0.f;
        for (int i = 0; i < n; ++i) if (!isnan(w[i])) c += w[i];
        return c;
    }
    if (stat == 1 || stat == 2 || stat == 3) {
        float mn = w[0], mx = w[0];
        for (int i = 1; i < n; ++i) {
            float v = w[i];
            if (!isnan(v)) {
                if (!(mn < v)) mn = v;
                if (!(mx > v)) mx = v;
            }
        }
        if (stat == 1) return mx;
        if (stat == 2) return mn;
        return mx - mn;
    }
    double c = 0.0; long cnt = 0;
    for (int i = 0; i < n; ++i) if (!isnan(w[i])) { c += (double)w[i]; ++cnt; }
    double m = c / (double)cnt;
    if (stat == 0) return (float)m;
    double ssd = 0.0;
    for (int i = 0; i < n; ++i) if (!isnan(w[i])) { double d = (double)w[i] - m; ssd += d * d; }
    double var = ssd / (double)cnt;
    if (stat == 5) return (float)var;
    return (float)pow(var, 0.5);
}

void orc_focal_apply(const float *data, float *out, int rows, int cols,
                     const double *kernel, int krows, int kcols, int stat, int nthreads) {
    int hrows = krows / 2, hcols = kcols / 2;
#pragma omp parallel num_threads(nthreads)
    {
        float *win = (float *)malloc(sizeof(float) * (size_t)krows * kcols);
#pragma omp for schedule(static)
        for (int y = 0; y < rows; ++y) {
            for (int x = 0; x < cols; ++x) {
                for (int i = 0; i < krows * kcols; ++i) win[i] = NAN;
                for (int ky = y - hrows; ky < y + hrows + 1; ++ky)
                    for (int kx = x - hcols; kx < x + hcols + 1; ++kx)
                        if (ky >= 0 && ky < rows && kx >= 0 && kx < cols) {
                            int kyi = ky - (y - hrows), kxi = kx - (x - hcols);
                            if (kernel[kyi * kcols + kxi] == 1)
                                win[kyi * kcols + kxi] = data[IDX(ky, kx)];
                        }
                out[IDX(y, x)] = reduce_window(win, krows * kcols, stat);
            }
        }
        free(win);
    }
}
