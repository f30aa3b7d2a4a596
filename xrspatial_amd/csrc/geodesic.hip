// Geodesic slope / aspect (method='geodesic'): WGS-84 ECEF -> local ENU plane fit per 3x3 window, float64.
//
// Reference: xrspatial/geodesic.py:40-229 (`_geodetic_to_ecef`, `_local_frame_project_and_fit`,
// `_geodesic_slope_at_point`, `_geodesic_aspect_at_point`, `_cpu_geodesic_*`), runners slope.py:167-174 and
// aspect.py:172-179.  Same sequence of operations as the reference (neighbours row-major, centred normal
// equations, |det| < 1e-30 -> flat); compute-bound (about 400 float64 flops per cell), not HBM-bound.
//
// Curvilinear grids (2-D lat / lon planes) follow the reference's sequence of operations (neighbours row-major, ECEF
// differences projected on the centre cell's frame, centred normal equations), with sincos per neighbour.
//
// Regular geographic grids (1-D lat / lon coordinates, the common case) never touch a trigonometric function per
// cell and never form an ECEF coordinate: on such a grid a neighbour's offset in the centre cell's East / North / Up
// frame depends only on its ROW (latitude terms) and on the longitude DIFFERENCE to the centre column,
//     R = (N + h) cos(lat),  Z = (b2/a2 N + h) sin(lat)                     (distance from the axis, height above the equator)
//     e = R_k sin(dlon),  q = R_k cos(dlon) - R_c,  dz = Z_k - Z_c
//     n = cos(lat_c) dz - sin(lat_c) q,   u = cos(lat_c) q + sin(lat_c) dz
// which is the reference's (P_k - P_c) . (east, north, up) with the centre longitude rotated out.  A prologue kernel
// tabulates sin, cos, N cos, (b2/a2) N sin per row and sin / cos of the two longitude steps per column; the cell
// kernel is ~14 float64 operations per neighbour instead of 23, the plane fit uses raw second moments (the centre
// offset is exactly 0) and the final atan / atan2 run in float32 (the result is float32).  Same float64 rounding
// level as the reference's own ECEF differences (both carry ~1e-9 m); the parity tests hold it to 1e-5 relative.
#include "xrs_common.h"

#include <cmath>

using namespace xrs;

namespace {

struct GeoArgs {
    const void *elev;           // float32 or float64 plane
    const double *lat, *lon;    // 1-D (lat[row], lon[col]) or 2-D planes with pitch ld_ll
    float *out;
    long rows, cols, ld_in, ld_out, ld_ll;
    int halo_top, halo_bot;
    int mode;                   // 0 slope, 1 aspect
    double a2, b2, zf, inv2r;
    const double *tab_lat;      // [rows + halos][4]: sin, cos, N cos, (b2/a2) N sin           (1-D grids)
    const double *tab_lon;      // [cols][4]: sin, cos of lon[x-1] - lon[x]; sin, cos of lon[x+1] - lon[x]
};

constexpr double kDeg2Rad = 3.141592653589793 / 180.0;
constexpr double kRad2Deg = 180.0 / 3.141592653589793;

struct LatT { double s, c, N, Nz; };
struct LonT { double s, c; };

__device__ __forceinline__ LatT lat_terms(double lat_deg, double a2, double b2) {
    LatT t;
    sincos(lat_deg * kDeg2Rad, &t.s, &t.c);
    t.N = a2 / sqrt(a2 * t.c * t.c + b2 * t.s * t.s);       // geodesic.py:47
    t.Nz = b2 / a2 * t.N;
    return t;
}
__device__ __forceinline__ LonT lon_terms(double lon_deg) {
    LonT t;
    sincos(lon_deg * kDeg2Rad, &t.s, &t.c);
    return t;
}

__global__ void geo_tables_kernel(const double *lat, long nlat, const double *lon, long nlon, double a2, double b2,
                                  double *tab_lat, double *tab_lon) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < nlat) {
        const LatT t = lat_terms(lat[i], a2, b2);
        tab_lat[4 * i] = t.s; tab_lat[4 * i + 1] = t.c; tab_lat[4 * i + 2] = t.N * t.c; tab_lat[4 * i + 3] = t.Nz * t.s;
    }
    if (i < nlon) {
        const LonT t = lon_terms(lon[i]);
        const LonT m = lon_terms(lon[i > 0 ? i - 1 : i]), p = lon_terms(lon[i + 1 < nlon ? i + 1 : i]);
        tab_lon[4 * i] = m.s * t.c - m.c * t.s;     tab_lon[4 * i + 1] = m.c * t.c + m.s * t.s;
        tab_lon[4 * i + 2] = p.s * t.c - p.c * t.s; tab_lon[4 * i + 3] = p.c * t.c + p.s * t.s;
    }
}

// slope (degrees) / aspect (compass degrees, -1 = flat) from the fitted plane u = A e + B n (geodesic.py:139-173)
__device__ __forceinline__ float plane_to_result(double A, double B, int mode) {
    const double mag2 = A * A + B * B;
    if (mode == 0) return atanf(__builtin_sqrtf((float)mag2)) * 57.29578f;
    if (mag2 < 1e-14) return -1.0f;
    double deg = (double)atan2f((float)-A, (float)-B) * kRad2Deg;
    if (deg < 0) deg += 360.0;
    if (deg >= 360.0) deg -= 360.0;
    return (float)deg;
}

// Regular grids: one cell per lane, a wave per row, 64 columns x 4 rows per workgroup.
template <typename ET>
__global__ void __launch_bounds__(256) geodesic_grid_kernel(const GeoArgs a, const long tiles_x, const long n_tiles) {
    const long tile = xcd_tile(blockIdx.x, n_tiles, XCD_UNIT(XRS_XCD_LDS, tiles_x));
    if (tile < 0) return;
    const long x = (tile % tiles_x) * 64 + (threadIdx.x & 63);
    const long y = (tile / tiles_x) * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (x >= a.cols || y >= a.rows) return;
    const long y_lo = -(long)a.halo_top, y_hi = a.rows + a.halo_bot;
    float res = nan_f32();
    const bool border = (y - 1 < y_lo) || (y + 1 >= y_hi) || x == 0 || x == a.cols - 1;
    if (!border) {
        const ET *elev = static_cast<const ET *>(a.elev);
        double h[9];
        bool ok = true;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const double v = (double)elev[(y + dy - 1) * a.ld_in + (x + dx - 1)];
                ok = ok && !isnan(v);
                h[dy * 3 + dx] = v * a.zf;
            }
        if (ok) {
            const double *tl = a.tab_lat + 4 * (y - 1 + a.halo_top);      // rows y-1, y, y+1 (wave-uniform)
            const double *to = a.tab_lon + 4 * x;
            const double sd[3] = {to[0], 0.0, to[2]}, cd[3] = {to[1], 1.0, to[3]};
            const double s0 = tl[4], c0 = tl[5];
            const double Rc = fma(h[4], c0, tl[6]), Zc = fma(h[4], s0, tl[7]);
            double Se = 0.0, Sn = 0.0, Su = 0.0, See = 0.0, Snn = 0.0, Sen = 0.0, Seu = 0.0, Snu = 0.0;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const double sr = tl[4 * dy], cr = tl[4 * dy + 1], Pr = tl[4 * dy + 2], Qr = tl[4 * dy + 3];
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    if (dy == 1 && dx == 1) continue;                       // the centre: offset exactly 0
                    const double hk = h[dy * 3 + dx];
                    const double R = fma(hk, cr, Pr), dz = fma(hk, sr, Qr) - Zc;
                    const double e = dx == 1 ? 0.0 : R * sd[dx];
                    const double q = dx == 1 ? R - Rc : fma(R, cd[dx], -Rc);
                    const double n = fma(c0, dz, -(s0 * q));
                    double u = fma(c0, q, s0 * dz);
                    u = fma(fma(e, e, n * n), a.inv2r, u);                  // curvature correction (geodesic.py:100-101)
                    Sn += n; Su += u;
                    Snn = fma(n, n, Snn); Snu = fma(n, u, Snu);
                    if (dx != 1) {
                        Se += e;
                        See = fma(e, e, See); Sen = fma(e, n, Sen); Seu = fma(e, u, Seu);
                    }
                }
            }
            // centred second moments from the raw ones: S_xy - (S_x S_y) / 9
            const double inv9 = 1.0 / 9.0;
            const double me = Se * inv9, mn = Sn * inv9;
            See = fma(-Se, me, See); Snn = fma(-Sn, mn, Snn); Sen = fma(-Se, mn, Sen);
            Seu = fma(-me, Su, Seu); Snu = fma(-mn, Su, Snu);
            const double det = See * Snn - Sen * Sen;
            double A = 0.0, B = 0.0;
            if (!(fabs(det) < 1e-30)) {
                const double inv = 1.0 / det;
                A = (Seu * Snn - Snu * Sen) * inv;
                B = (Snu * See - Seu * Sen) * inv;
            }
            res = plane_to_result(A, B, a.mode);
        }
    }
    st_stream(&a.out[y * a.ld_out + x], res);
}

// Curvilinear grids: the reference's own sequence, per-neighbour trigonometry.
template <typename ET>
__global__ void __launch_bounds__(256) geodesic_kernel(const GeoArgs a, const long tiles_x, const long n_tiles) {
    // one workgroup = 64 columns x 4 rows (a wave per row); tiles numbered row-major, dealt to the XCDs in bands
    const long tile = xcd_tile(blockIdx.x, n_tiles, XCD_UNIT(XRS_XCD_LDS, tiles_x));
    if (tile < 0) return;
    const long x = (tile % tiles_x) * 64 + (threadIdx.x & 63);
    const long y = (tile / tiles_x) * 4 + (threadIdx.x >> 6);
    if (x >= a.cols || y >= a.rows) return;
    const long y_lo = -(long)a.halo_top, y_hi = a.rows + a.halo_bot;
    float res = nan_f32();
    const bool border = (y - 1 < y_lo) || (y + 1 >= y_hi) || x == 0 || x == a.cols - 1;
    if (!border) {
        const ET *elev = static_cast<const ET *>(a.elev);
        double h[9];
        bool ok = true;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const double v = (double)elev[(y + dy - 1) * a.ld_in + (x + dx - 1)];
                ok = ok && !isnan(v);
                h[dy * 3 + dx] = v * a.zf;
            }
        if (ok) {
            LatT la[9];
            LonT lo[9];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int k = dy * 3 + dx;
                    la[k] = lat_terms(a.lat[(y + dy - 1) * a.ld_ll + (x + dx - 1)], a.a2, a.b2);
                    lo[k] = lon_terms(a.lon[(y + dy - 1) * a.ld_ll + (x + dx - 1)]);
                }
            // centre cell: ECEF and local East / North / Up unit vectors (geodesic.py:71-82)
            const LatT lc = la[4];
            const LonT oc = lo[4];
            const double Xc = (lc.N + h[4]) * lc.c * oc.c, Yc = (lc.N + h[4]) * lc.c * oc.s, Zc = (lc.Nz + h[4]) * lc.s;
            const double ex = -oc.s, ey = oc.c;
            const double nx = -lc.s * oc.c, ny = -lc.s * oc.s, nz = lc.c;
            const double ux = lc.c * oc.c, uy = lc.c * oc.s, uz = lc.s;
            double e9[9], n9[9], u9[9];
            double me = 0.0, mn = 0.0, mu = 0.0;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                if (k == 4) {            // the centre itself: (P - P) projected on any axis is exactly 0
                    e9[k] = 0.0; n9[k] = 0.0; u9[k] = 0.0;
                    continue;
                }
                const double Xk = (la[k].N + h[k]) * la[k].c * lo[k].c;
                const double Yk = (la[k].N + h[k]) * la[k].c * lo[k].s;
                const double Zk = (la[k].Nz + h[k]) * la[k].s;
                const double ddx = Xk - Xc, ddy = Yk - Yc, ddz = Zk - Zc;
                const double ek = ddx * ex + ddy * ey;
                const double nk = ddx * nx + ddy * ny + ddz * nz;
                double uk = ddx * ux + ddy * uy + ddz * uz;
                uk += (ek * ek + nk * nk) * a.inv2r;           // curvature correction (geodesic.py:100-101)
                e9[k] = ek; n9[k] = nk; u9[k] = uk;
                me += ek; mn += nk; mu += uk;
            }
            const double inv9 = 1.0 / 9.0;
            me *= inv9; mn *= inv9; mu *= inv9;
            double See = 0.0, Snn = 0.0, Sen = 0.0, Seu = 0.0, Snu = 0.0;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const double de = e9[k] - me, dn = n9[k] - mn, du = u9[k] - mu;
                See += de * de; Snn += dn * dn; Sen += de * dn; Seu += de * du; Snu += dn * du;
            }
            const double det = See * Snn - Sen * Sen;
            double A = 0.0, B = 0.0;
            if (!(fabs(det) < 1e-30)) {
                A = (Seu * Snn - Snu * Sen) / det;
                B = (Snu * See - Seu * Sen) / det;
            }
            const double mag = sqrt(A * A + B * B);
            if (a.mode == 0) {
                res = (float)(atan(mag) * kRad2Deg);
            } else if (mag < 1e-7) {
                res = -1.0f;
            } else {
                double deg = atan2(-A, -B) * kRad2Deg;
                if (deg < 0) deg += 360.0;
                if (deg >= 360.0) deg -= 360.0;
                res = (float)deg;
            }
        }
    }
    st_stream(&a.out[y * a.ld_out + x], res);
}

}  // namespace

extern "C" {

size_t xrs_geodesic_workspace_bytes(int64_t rows_with_halos, int64_t cols) {
    if (rows_with_halos < 0 || cols < 0) return 0;
    return (size_t)(4 * rows_with_halos + 4 * cols) * sizeof(double) + 64;
}

int xrs_geodesic_f32(const void *elev_dev, int elev_is_f64, const double *lat_dev, const double *lon_dev, int latlon_2d,
                     float *out_dev, int64_t rows, int64_t cols, int64_t ld_in, int64_t ld_out, int64_t ld_latlon,
                     double a2, double b2, double z_factor, int aspect, void *work_dev, int halo_top, int halo_bot,
                     void *stream) {
    if (!elev_dev || !lat_dev || !lon_dev || !out_dev) return fail("xrs_geodesic_f32: null pointer");
    if (rows < 0 || cols < 0 || ld_in < cols || ld_out < cols || halo_top < 0 || halo_bot < 0)
        return fail("xrs_geodesic_f32: bad shape");
    if (latlon_2d && ld_latlon < cols) return fail("xrs_geodesic_f32: bad lat/lon pitch");
    if (!latlon_2d && !work_dev) return fail("xrs_geodesic_f32: 1-D coordinates need a workspace");
    if (rows == 0 || cols == 0) return 0;
    GeoArgs a;
    memset(&a, 0, sizeof(a));
    a.elev = elev_dev; a.lat = lat_dev; a.lon = lon_dev; a.out = out_dev;
    a.rows = rows; a.cols = cols; a.ld_in = ld_in; a.ld_out = ld_out; a.ld_ll = ld_latlon;
    a.halo_top = halo_top; a.halo_bot = halo_bot; a.mode = aspect ? 1 : 0;
    a.a2 = a2; a.b2 = b2; a.zf = z_factor;
    a.inv2r = 1.0 / (2.0 * 6370994.884953014);                  // WGS84 mean radius (geodesic.py:187)
    hipStream_t s = as_stream(stream);
    const long tiles_x = (cols + 63) / 64, n_tiles = tiles_x * ((rows + 3) / 4);
    const dim3 grid((unsigned)xcd_grid(n_tiles, XCD_UNIT(XRS_XCD_LDS, tiles_x)));
    if (!latlon_2d) {
        const long nlat = rows + halo_top + halo_bot;
        double *tab_lat = static_cast<double *>(work_dev);
        double *tab_lon = tab_lat + 4 * nlat;
        const long nmax = nlat > cols ? nlat : cols;
        hipLaunchKernelGGL(geo_tables_kernel, dim3((unsigned)((nmax + 255) / 256)), dim3(256), 0, s,
                           lat_dev - halo_top, nlat, lon_dev, (long)cols, a2, b2, tab_lat, tab_lon);
        a.tab_lat = tab_lat; a.tab_lon = tab_lon;
        if (elev_is_f64) hipLaunchKernelGGL(geodesic_grid_kernel<double>, grid, dim3(256), 0, s, a, tiles_x, n_tiles);
        else hipLaunchKernelGGL(geodesic_grid_kernel<float>, grid, dim3(256), 0, s, a, tiles_x, n_tiles);
    } else {
        if (elev_is_f64) hipLaunchKernelGGL(geodesic_kernel<double>, grid, dim3(256), 0, s, a, tiles_x, n_tiles);
        else hipLaunchKernelGGL(geodesic_kernel<float>, grid, dim3(256), 0, s, a, tiles_x, n_tiles);
    }
    XRS_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
