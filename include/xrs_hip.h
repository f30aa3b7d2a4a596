/*
 * xrs_hip.h -- C ABI of libxrs_hip.so, the MI355X (gfx950) backend for the dense 2-D
 * raster hot path of xarray-spatial.
 *
 * The reference has no FFI of its own: its only extension seam is the four-slot
 * backend table ArrayTypeFunctionMapping(numpy_func, cupy_func, dask_func,
 * dask_cupy_func) (xrspatial/utils.py:117-143) whose slots are per-backend
 * "runner" functions taking raw arrays.  Each entry point below replaces one such
 * runner (cited per function); the Python host layer (xrspatial_amd/) binds them
 * with ctypes and keeps validation / metadata / Dataset handling above the ABI.
 *
 * Conventions
 *  - every function returns 0 on success, non-zero on failure; the message of the
 *    calling thread's last failure is read with xrs_last_error().
 *  - plain pointers and sizes only.  Pointers named *_dev are device (HBM)
 *    pointers obtained from xrs_malloc(); everything else is host memory.
 *  - the caller owns every buffer; the library never allocates outputs, never
 *    frees inputs, and keeps no global mutable state (re-entrant: the reference's
 *    Numba kernels are nogil and may be called from several threads).
 *  - rasters are C-order float32 planes: element (y, x) of a plane `p` with row
 *    pitch `ld` (in ELEMENTS) is p[y * ld + x].  `rows` counts the rows this call
 *    OWNS (and writes); `halo_top` / `halo_bot` say how many extra valid input
 *    rows sit directly above `in_dev` (at negative row indices) / below row
 *    rows-1.  0 means that side is a true raster edge.  This is the row-shard
 *    contract of the multi-GPU path (reference semantics: dask
 *    map_overlap(depth=k//2, boundary=nan), e.g. xrspatial/slope.py:94-97).
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All
 *    kernels are asynchronous on it.
 */
#ifndef XRS_HIP_H
#define XRS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ runtime */
int xrs_version(void);                                   /* ABI version, currently 1 */
int xrs_last_error(char *buf, size_t buflen);            /* copies the thread's last error text */
/* 16 hex digits identifying the SOURCES this library was built from (sha256 of csrc/ + this header, made by the
 * Makefile): profiles/pmc_traffic.json records the build its counters were collected on and bench.py reports them only
 * for that build. */
int xrs_build_id(char *buf, size_t buflen);
int xrs_device_count(int *count);
int xrs_set_device(int device);
int xrs_get_device(int *device);
int xrs_device_name(int device, char *buf, size_t buflen);
int xrs_mem_info(size_t *free_bytes, size_t *total_bytes);
int xrs_malloc(void **ptr_dev, size_t bytes);
int xrs_free(void *ptr_dev);
/* Page-locked host memory: copies to / from it run at PCIe rate and are truly asynchronous on `stream`
 * (the host layer stages numpy-backed inputs / results through a pool of such blocks). */
int xrs_host_alloc(void **ptr_host, size_t bytes);
int xrs_host_free(void *ptr_host);
int xrs_memcpy_h2d(void *dst_dev, const void *src, size_t bytes, void *stream);
int xrs_memcpy_d2h(void *dst, const void *src_dev, size_t bytes, void *stream);
int xrs_memcpy_d2d(void *dst_dev, const void *src_dev, size_t bytes, void *stream);
int xrs_memset(void *dst_dev, int byte_value, size_t bytes, void *stream);
/* Calibration: the streaming pattern of xrs_copy_f32 with ONE plane read and n_dst (1..8) planes written -- the ceiling
 * of a fused kernel's own read/write mix (HBM3E sustains less on write-heavy mixes than on a 1:1 copy).  `dsts_dev` is a
 * HOST array of device pointers.  Used by bench.py / tools/kbench.py next to xrs_copy_f32; not on the data path. */
int xrs_stream_mix_f32(const float *src_dev, float *const *dsts_dev, int n_dst, int64_t n, void *stream);
/* a rectangular window of a plane (pitches and width in bytes): `raster[top:bottom + 1, left:right + 1]` of a
 * device-resident raster (the slicing at the end of zonal.trim / zonal.crop, xrspatial/zonal.py:1841, 2058) */
int xrs_copy2d(void *dst_dev, size_t dst_pitch, const void *src_dev, size_t src_pitch, size_t width_bytes,
               int64_t rows, void *stream);
/* streaming plane copy in the library's own access pattern: the measured-copy-bandwidth calibration point */
/* `.astype(np.float32)` of the reference's wrappers (xrspatial/slope.py:82, hillshade.py:21, multispectral.py:834 ...)
 * on the device: numpy-backed rasters are sent in their own dtype and converted in HBM (round to nearest even). */
enum { XRS_DT_I8 = 0, XRS_DT_U8 = 1, XRS_DT_I16 = 2, XRS_DT_U16 = 3, XRS_DT_I32 = 4, XRS_DT_U32 = 5,
       XRS_DT_I64 = 6, XRS_DT_U64 = 7, XRS_DT_F64 = 8, XRS_DT_F32 = 9 };
int xrs_cast_f32(const void *src_dev, int src_dtype, float *dst_dev, int64_t n, void *stream);
int xrs_copy_f32(const float *src_dev, float *dst_dev, int64_t n, void *stream);
int xrs_stream_create(void **stream);
int xrs_stream_destroy(void *stream);
int xrs_stream_sync(void *stream);
int xrs_device_sync(void);
int xrs_event_create(void **event);
int xrs_event_destroy(void *event);
int xrs_event_record(void *event, void *stream);
int xrs_stream_wait_event(void *stream, void *event);   /* later work on `stream` waits for `event` (device side) */
int xrs_event_sync(void *event);
int xrs_event_elapsed_ms(void *start_event, void *stop_event, float *ms);

/* --------------------------------------------------------- 3x3 terrain family
 * One-cell NaN border on true raster edges.  Replace the runners
 *   slope      xrspatial/slope.py:79-83      (_run_numpy -> _cpu :56-76)
 *   aspect     xrspatial/aspect.py:56-90     (_run_numpy)
 *   curvature  xrspatial/curvature.py:44-49  (_run_numpy -> _cpu :31-41)
 *   hillshade  xrspatial/hillshade.py:20-35  (_run_numpy)
 * Arithmetic follows the reference's CPU path (float64 gradient sums), not its
 * all-float32 CuPy kernels.  Outputs are float32; hillshade can also write
 * float64 (out_f64 != 0), which is what the reference returns under NumPy >= 2. */
int xrs_slope_f32(const float *in_dev, float *out_dev, int64_t rows, int64_t cols,
                  int64_t ld_in, int64_t ld_out, double cellsize_x, double cellsize_y,
                  int halo_top, int halo_bot, void *stream);
int xrs_aspect_f32(const float *in_dev, float *out_dev, int64_t rows, int64_t cols,
                   int64_t ld_in, int64_t ld_out, int halo_top, int halo_bot, void *stream);
int xrs_curvature_f32(const float *in_dev, float *out_dev, int64_t rows, int64_t cols,
                      int64_t ld_in, int64_t ld_out, double cellsize,
                      int halo_top, int halo_bot, void *stream);
int xrs_hillshade_f32(const float *in_dev, void *out_dev, int out_f64, int64_t rows, int64_t cols,
                      int64_t ld_in, int64_t ld_out, double azimuth, double angle_altitude,
                      int halo_top, int halo_bot, void *stream);

/* Fused variant: one read of the DEM, up to four outputs.  Any of the output
 * pointers may be NULL (that product is skipped).  Same results as the four
 * separate calls.  (The reference's summarize_terrain, xrspatial/analytics.py,
 * makes three separate passes.) */
int xrs_terrain_fused_f32(const float *in_dev, float *slope_dev, float *aspect_dev,
                          float *curvature_dev, float *hillshade_dev,
                          int64_t rows, int64_t cols, int64_t ld_in, int64_t ld_out,
                          double cellsize_x, double cellsize_y, double azimuth, double angle_altitude,
                          int halo_top, int halo_bot, void *stream);

/* Fused raster pass: ONE read of the raster, any subset of the four terrain products plus a focal mean
 * (focal.apply / focal_stats 'mean' with a 0/1 mask).  Replaces a sequence of separate reference passes over
 * the same DataArray -- hillshade (xrspatial/hillshade.py:20-35), slope (slope.py:56-76), aspect
 * (aspect.py:56-90), curvature (curvature.py:31-49), focal.apply with _calc_mean (focal.py:226-228, 305-326)
 * -- and returns bit-identical results to the separate entry points above / xrs_focal_stats_f32.
 * Any output pointer may be NULL.  3x3 and 5x5 masks on 16-byte-friendly rasters run as one kernel
 * (4 B read + 4 B written per product per cell); other shapes fall back to the separate launches
 * (`work_dev`: as for xrs_focal_stats_f32, may be NULL for masks up to 5x5).
 * Row shards: halo_top / halo_bot must be 0 (true raster edge) or >= max(1, krows/2). */
int xrs_raster_pass_f32(const float *in_dev, float *slope_dev, float *aspect_dev, float *curvature_dev,
                        float *hillshade_dev, float *focal_mean_dev, const double *kernel, int krows,
                        int kcols, void *work_dev, int64_t rows, int64_t cols, int64_t ld_in, int64_t ld_out,
                        double cellsize_x, double cellsize_y, double azimuth, double angle_altitude,
                        int halo_top, int halo_bot, void *stream);

/* The same pass over only the first and the last `edge_rows` rows of the raster (a row shard whose halo rows have just
 * arrived: the rows in between were launched earlier, while the exchange was in flight -- what a dask graph does with
 * map_overlap(depth, boundary=nan), slope.py:86-97, the scheduler does here with two streams).  ONE launch over two
 * segments of tile rows when the fused kernel takes the request (3x3 / 5x5 masks: results are those of xrs_raster_pass_f32
 * on the whole raster, bit for bit), the two sub-range calls otherwise (what any row split of that request gives).  The
 * one-launch form works in whole tile rows of 16 raster rows: besides the edges it also writes up to 15 rows BELOW the first
 * edge (rows [edge_rows, 16 * ceil(edge_rows / 16))) and up to 15 rows ABOVE the last one, with the values the whole-raster
 * pass gives them -- harmless on the stream that also runs the launch for the rows in between (same values), a write-write
 * race if that launch runs on another stream: keep both on one stream, or pass edge_rows % 16 == 0.  No other row between the
 * edges is written.  2 * edge_rows >= rows: the whole raster. */
int xrs_raster_pass_edges_f32(const float *in_dev, float *slope_dev, float *aspect_dev, float *curvature_dev,
                              float *hillshade_dev, float *focal_mean_dev, const double *kernel, int krows,
                              int kcols, void *work_dev, int64_t rows, int64_t cols, int64_t ld_in, int64_t ld_out,
                              double cellsize_x, double cellsize_y, double azimuth, double angle_altitude,
                              int halo_top, int halo_bot, int64_t edge_rows, void *stream);

/* Geodesic slope / aspect (method='geodesic'): WGS-84 ECEF -> local ENU plane fit per 3x3 window, float64
 * arithmetic, float32 out, NaN border, NaN if any of the nine elevations is NaN.  Replaces
 * _cpu_geodesic_slope / _cpu_geodesic_aspect (xrspatial/geodesic.py:181-229) behind slope.py:167-174 and
 * aspect.py:172-179.  elev_is_f64 selects the elevation element type (the reference widens to float64;
 * float32 rasters are widened in registers).  latlon_2d = 0: lat_dev[row] / lon_dev[col] are 1-D degree
 * coordinates (lat_dev points at the first OWNED row; halo rows' latitudes precede / follow it) and
 * `work_dev` must hold xrs_geodesic_workspace_bytes(rows + halo_top + halo_bot, cols) bytes for the
 * per-row / per-column trigonometry tables.  latlon_2d = 1: 2-D planes with pitch ld_latlon, no workspace.
 * aspect = 0 -> slope in degrees; 1 -> compass aspect, -1 where the fitted gradient is below 1e-7. */
size_t xrs_geodesic_workspace_bytes(int64_t rows_with_halos, int64_t cols);
int xrs_geodesic_f32(const void *elev_dev, int elev_is_f64, const double *lat_dev, const double *lon_dev,
                     int latlon_2d, float *out_dev, int64_t rows, int64_t cols, int64_t ld_in, int64_t ld_out,
                     int64_t ld_latlon, double a2, double b2, double z_factor, int aspect, void *work_dev,
                     int halo_top, int halo_bot, void *stream);

/* ----------------------------------------------------------- per-cell indices
 * Flat arrays of n float32 cells, NaN where the denominator is exactly 0.
 *   normalized_ratio  xrspatial/multispectral.py:825-841 (_normalized_ratio_cpu: ndvi/nbr/nbr2/ndmi)
 *   evi               xrspatial/multispectral.py:175-188 (_evi_cpu)
 *   savi              xrspatial/multispectral.py:876-890 (_savi_cpu) */
int xrs_normalized_ratio_f32(const float *a_dev, const float *b_dev, float *out_dev, int64_t n, void *stream);
int xrs_evi_f32(const float *nir_dev, const float *red_dev, const float *blue_dev, float *out_dev,
                int64_t n, double c1, double c2, double soil_factor, double gain, void *stream);
int xrs_savi_f32(const float *nir_dev, const float *red_dev, float *out_dev, int64_t n,
                 double soil_factor, void *stream);
/*   arvi  xrspatial/multispectral.py:29-43     gci   :350-361
 *   sipi  xrspatial/multispectral.py:1017-1031 ebbi  :1160-1174 */
int xrs_arvi_f32(const float *nir_dev, const float *red_dev, const float *blue_dev, float *out_dev,
                 int64_t n, void *stream);
int xrs_gci_f32(const float *nir_dev, const float *green_dev, float *out_dev, int64_t n, void *stream);
int xrs_sipi_f32(const float *nir_dev, const float *red_dev, const float *blue_dev, float *out_dev,
                 int64_t n, void *stream);
int xrs_ebbi_f32(const float *red_dev, const float *swir_dev, const float *tir_dev, float *out_dev,
                 int64_t n, void *stream);

/* ------------------------------------------------------------- k x k kernels
 * convolve2d: correlation with a float64 weight matrix given on the HOST
 * (krows x kcols, both odd, C order); NaN border of k//2 on true edges, NaNs
 * propagate.  Replaces _convolve_2d_numpy, xrspatial/convolution.py:285-313.
 * `work_dev` must hold xrs_kxk_workspace_bytes(krows, kcols) bytes (the weights
 * are staged there on `stream`). */
size_t xrs_kxk_workspace_bytes(int krows, int kcols);
int xrs_convolve2d_f32(const float *in_dev, float *out_dev, int64_t rows, int64_t cols,
                       int64_t ld_in, int64_t ld_out, const double *kernel, int krows, int kcols,
                       void *work_dev, int halo_top, int halo_bot, void *stream);

/* focal statistics over the window cells where kernel == 1 exactly, window clipped
 * to the raster (no NaN border), NaN cells skipped; all requested statistics in ONE
 * pass.  Replaces _apply_numpy with the built-in reducers, xrspatial/focal.py:305-326
 * and :268-302, i.e. focal.apply(func=_calc_*) and focal.focal_stats (:782-797).
 * stat_mask bit i selects XRS_STAT_*; outs_dev[i] (HOST array of 7 device
 * pointers) receives statistic i and may be NULL when its bit is clear. */
enum { XRS_STAT_MEAN = 0, XRS_STAT_MAX = 1, XRS_STAT_MIN = 2, XRS_STAT_RANGE = 3,
       XRS_STAT_STD = 4, XRS_STAT_VAR = 5, XRS_STAT_SUM = 6, XRS_NUM_STATS = 7 };
int xrs_focal_stats_f32(const float *in_dev, float *const *outs_dev, unsigned stat_mask,
                        int64_t rows, int64_t cols, int64_t ld_in, int64_t ld_out,
                        const double *kernel, int krows, int kcols, void *work_dev,
                        int halo_top, int halo_bot, void *stream);
/* The same with accuracy options (xrs_focal_stats_f32 == flags 0).  Windows of 9x9 cells and more take float32 walkers
 * whose mean / var / std / sum agree with the reference's float64 accumulators to <= 2e-6 relative (a guard per output
 * sends ill-conditioned tiles to the exact kernels); the flags keep whole launches on the exact kernels instead:
 *   XRS_FOCAL_EXACT_MOMENTS   mean / var / std from float64 running sums (NaN-skipping, counted): ~1 ulp of the reference;
 *   XRS_FOCAL_SEQUENTIAL_SUM  `sum` added tap by tap in the reference's row-major order in float32 (numba's nansum keeps
 *                             the array dtype): bit-identical to the CPU path.
 * The host layer sets them from xrspatial_amd.focal.options / the XRS_FOCAL_SUM and XRS_FOCAL_MOMENTS variables. */
enum { XRS_FOCAL_EXACT_MOMENTS = 1, XRS_FOCAL_SEQUENTIAL_SUM = 2 };
int xrs_focal_stats_f32_ex(const float *in_dev, float *const *outs_dev, unsigned stat_mask,
                           int64_t rows, int64_t cols, int64_t ld_in, int64_t ld_out,
                           const double *kernel, int krows, int kcols, void *work_dev, size_t work_bytes,
                           int halo_top, int halo_bot, unsigned flags, void *stream);
/* Device scratch xrs_focal_stats_f32_ex can use for a rows x cols raster: the float64 copy of the kernel
 * (xrs_kxk_workspace_bytes) plus one byte per tile for the separable box kernel -- np.ones((k, k)) masks, the ones the
 * reference's benchmark suite runs, get their variance / standard deviation (and the mean and sum beside them) from an
 * O(1)-per-cell walk that hands the tiles it cannot stand for (NaN / inf cells, flat windows) to the general kernel
 * through that map.  With a smaller workspace (or none) the general kernel runs everywhere: more time, and results that agree
 * with the separable walk's within the documented tolerance (float32 walker vs float64 column sums: ~2e-6 relative on var / std),
 * not bit for bit. */
size_t xrs_focal_workspace_bytes(int64_t rows, int64_t cols, int krows, int kcols);

/* focal.apply with a user callable (func other than the built-in reducers): the kernel-shaped float32 arrays that
 * _apply_numpy, xrspatial/focal.py:305-326, builds per cell (NaN, then data[ky, kx] where kernel == 1 and inside the
 * raster), for the band of rows [y0, y0 + band_rows): windows_dev[band_rows][cols][krows][kcols].  The host calls the
 * callable on each.  in_dev is the whole raster (rows x cols, pitch ld_in); krows * kcols <= 2048. */
int xrs_focal_windows_f32(const float *in_dev, float *windows_dev, int64_t rows, int64_t cols, int64_t ld_in,
                          int64_t y0, int64_t band_rows, const double *kernel, int krows, int kcols, void *stream);

/* focal.mean: fixed 3x3 NaN-skipping mean on a clamped window, float64 out, cells
 * equal to one of `excludes` (NaN matches NaN) are passed through.  One pass;
 * the host loops `passes`.  Replaces _mean_numpy, xrspatial/focal.py:44-67.
 * in_is_f64 selects the input element type (first pass may read float32). */
int xrs_focal_mean3x3(const void *in_dev, int in_is_f64, double *out_dev, int64_t rows, int64_t cols,
                      int64_t ld_in, int64_t ld_out, const double *excludes, int n_excludes,
                      int halo_top, int halo_bot, void *stream);

/* `passes` applications in one call (the loop of focal.py:257-259): contiguous planes (pitch = cols), the passes
 * ping-pong between out_dev and scratch_dev (rows*cols doubles, may be NULL for passes == 1); result in out_dev. */
int xrs_focal_mean3x3_passes(const void *in_dev, int in_is_f64, double *out_dev, double *scratch_dev, int passes,
                             int64_t rows, int64_t cols, const double *excludes, int n_excludes, void *stream);

/* focal.hotspots support (xrspatial/focal.py:881-934): global NaN-skipping moments of a float32 plane
 * -> moments32_dev = { uint64 count; double sum, ssd (sum of squared deviations from the mean), mean },
 * and the z-score classifier  z = (mean_array - global_mean) / global_std  ->  {0, +-90, +-95, +-99} int8. */
int xrs_nan_moments_f32(const float *in_dev, int64_t n, void *moments32_dev, void *stream);
int xrs_hotspots_classify_f32(const float *mean_array_dev, signed char *out_dev, int64_t n,
                              float global_mean, float global_std, void *stream);

/* -------------------------------------------------------------------- zonal
 * Per-zone partial reductions of one streaming pass over (zone index, value):
 * count (integer-exact), sum and sum of squares (float64), min, max.  Cells with
 * zone index < 0 or >= n_zones, non-finite values, and values == nodata (when
 * has_nodata) are skipped.  The five output arrays (n_zones entries each) are
 * ACCUMULATED into: initialise them with xrs_zonal_init().  sum / sumsq hold the sums of
 * (x - shift) and (x - shift)^2 for the caller's `shift` (any value near the data: mean = shift + sum / n,
 * var = (sumsq - sum^2 / n) / n then stays well conditioned for rasters with a large offset and a small
 * spread; 0 gives the plain sums).  These partials are
 * the per-block statistics of the reference's dask path (_DASK_BLOCK_STATS,
 * xrspatial/zonal.py:83-89) from which mean/std/var follow (:100-102); the
 * NumPy path they replace is _stats_numpy (:280-332). */
int xrs_zonal_init(uint64_t *count_dev, double *sum_dev, double *sumsq_dev,
                   float *min_dev, float *max_dev, int n_zones, void *stream);
int xrs_zonal_partials_f32(const int32_t *zone_idx_dev, const float *values_dev, int64_t n,
                           int n_zones, float nodata, int has_nodata, double shift,
                           uint64_t *count_dev, double *sum_dev, double *sumsq_dev,
                           float *min_dev, float *max_dev, void *stream);
/* Same reduction straight from the RAW int32 zone raster: `lut_dev[id - zone_min]` (zone_range entries, built from
 * xrs_zonal_scan / xrs_zonal_presence) gives the dense index of an id, -1 for ids that are not a zone; ids outside the
 * window belong to no zone.  Saves materialising the dense index raster (4 B written + 4 B read per cell) when only
 * the partial-sum statistics are wanted (np.unique(zones) + per-zone masks in the reference, zonal.py:290-311). */
int xrs_zonal_partials_lut_f32(const int32_t *zones_dev, int32_t zone_min, int32_t zone_range, const int32_t *lut_dev,
                               const float *values_dev, int64_t n, int n_zones, float nodata, int has_nodata, double shift,
                               uint64_t *count_dev, double *sum_dev, double *sumsq_dev, float *min_dev, float *max_dev,
                               void *stream);
int xrs_zonal_partials_lut_f64(const int32_t *zones_dev, int32_t zone_min, int32_t zone_range, const int32_t *lut_dev,
                               const double *values_dev, int64_t n, int n_zones, double nodata, int has_nodata, double shift,
                               uint64_t *count_dev, double *sum_dev, double *sumsq_dev, double *min_dev, double *max_dev,
                               void *stream);
/* ONE pass from the raw int32 zone raster WITHOUT a discovery pass in front (np.unique(zones) of zonal.py:290 folded into the
 * reduction): the caller GUESSES a window of ids [zone_base, zone_base + window) -- xrs_zonal_sample_* below reads a strided
 * sample of the rasters for that -- and the kernel accumulates straight into window-indexed tables (entry id - zone_base;
 * window <= 5000 ids: the tables live in LDS), which it initialises itself (overwritten, not accumulated into).
 * present_dev[id - zone_base] = 1 (window bytes) for ids that occur with invalid values only (count 0: the reference still
 * lists such a zone); *overflow_dev = 1 if some cell's id lies outside the window -- the tables are then incomplete and the
 * caller takes the two-pass route (xrs_zonal_scan_presence_i32 + xrs_zonal_partials_lut_*). */
int xrs_zonal_partials_window_f32(const int32_t *zones_dev, int32_t zone_base, int window, const float *values_dev, int64_t n,
                                  float nodata, int has_nodata, double shift, uint64_t *count_dev, double *sum_dev,
                                  double *sumsq_dev, float *min_dev, float *max_dev, unsigned char *present_dev,
                                  int32_t *overflow_dev, void *stream);
int xrs_zonal_partials_window_f64(const int32_t *zones_dev, int32_t zone_base, int window, const double *values_dev, int64_t n,
                                  double nodata, int has_nodata, double shift, uint64_t *count_dev, double *sum_dev,
                                  double *sumsq_dev, double *min_dev, double *max_dev, unsigned char *present_dev,
                                  int32_t *overflow_dev, void *stream);
/* n_samples cells at an odd stride through both rasters -> result24_dev = { int32 zmin, zmax; double SUM of the valid
 * values; uint64 number of valid values }: the id window to guess and (sum / count) a shift for the moments, from two tiny
 * launches. */
int xrs_zonal_sample_f32(const int32_t *zones_dev, const float *values_dev, int64_t n, int64_t n_samples, float nodata,
                         int has_nodata, void *result24_dev, void *stream);
int xrs_zonal_sample_f64(const int32_t *zones_dev, const double *values_dev, int64_t n, int64_t n_samples, double nodata,
                         int has_nodata, void *result24_dev, void *stream);
/* float64 values (the reference does not cast `values`: float64 and integer rasters keep
 * their precision; integers are widened to float64 by the host layer).  min/max are float64. */
int xrs_zonal_init_f64(uint64_t *count_dev, double *sum_dev, double *sumsq_dev,
                       double *min_dev, double *max_dev, int n_zones, void *stream);
int xrs_zonal_partials_f64(const int32_t *zone_idx_dev, const double *values_dev, int64_t n,
                           int n_zones, double nodata, int has_nodata, double shift,
                           uint64_t *count_dev, double *sum_dev, double *sumsq_dev,
                           double *min_dev, double *max_dev, void *stream);

/* Dense zone indexing on the device (replaces the host-side np.unique of xrspatial/zonal.py:290 for
 * integral zone ids): zone_dtype 0 = int32, 1 = int64, 2 = float32, 3 = float64.
 *   xrs_zonal_scan      -> result32_dev = { double zmin, zmax; uint64 n_finite; int32 all_integral, pad }
 *   xrs_zonal_presence  -> present_dev[id - zmin] = 1 for every finite id in [zmin, zmin + range)
 *   xrs_zonal_index     -> idx_dev[cell] = lut_dev[id - zmin], -1 for non-finite / out-of-range ids */
int xrs_zonal_scan(const void *zones_dev, int zone_dtype, int64_t n, void *result32_dev, void *stream);
/* scan + presence in ONE read for int32 ids: ids inside the optimistic window [0, window) are marked in present_dev
 * (window bytes) during the scan; if the scan result shows 0 <= zmin and zmax < window the map is complete and
 * xrs_zonal_presence is not needed. */
int xrs_zonal_scan_presence_i32(const int32_t *zones_dev, int64_t n, void *result32_dev, unsigned char *present_dev,
                                int window, void *stream);
int xrs_zonal_presence(const void *zones_dev, int zone_dtype, int64_t n, double zmin, int64_t range,
                       unsigned char *present_dev, void *stream);
int xrs_zonal_index(const void *zones_dev, int zone_dtype, int64_t n, double zmin, int64_t range,
                    const int32_t *lut_dev, int32_t *idx_dev, void *stream);

/* zonal.crosstab, 2-D values (xrspatial/zonal.py:699-800): counts_dev[zone * n_cats + cat] += number of cells
 * with that (dense zone index, dense category index) pair; negative / out-of-range indices are skipped.
 * counts_dev (n_zones * n_cats uint64) is accumulated into: zero it with xrs_memset first. */
int xrs_crosstab_counts(const int32_t *zone_idx_dev, const int32_t *cat_idx_dev, int64_t n, int n_zones,
                        int n_cats, uint64_t *counts_dev, void *stream);

/* majority (most frequent valid value per zone, ties -> smallest; NaN for zones without a valid
 * cell), computed by two device radix sorts + run voting; replaces _stats_majority applied per zone
 * (xrspatial/zonal.py:56-68, 144-163).  `work_dev` must hold
 * xrs_zonal_majority_workspace_bytes(n, n_zones, values_f64) bytes.  n < 2^31 cells per call. */
size_t xrs_zonal_majority_workspace_bytes(int64_t n, int n_zones, int values_f64);
int xrs_zonal_majority_f32(const int32_t *zone_idx_dev, const float *values_dev, int64_t n, int n_zones,
                           float nodata, int has_nodata, void *work_dev, size_t work_bytes,
                           double *majority_dev, void *stream);
int xrs_zonal_majority_f64(const int32_t *zone_idx_dev, const double *values_dev, int64_t n, int n_zones,
                           double nodata, int has_nodata, void *work_dev, size_t work_bytes,
                           double *majority_dev, void *stream);
/* majority WITHOUT a sort (csrc/zonal_mode.hip): the cells are routed zone by zone and then, inside a zone, by a hash
 * of the value's bits until every part fits an LDS hash table that counts multiplicities; same result as
 * xrs_zonal_majority_* (ties -> smallest value, -0.0 == +0.0, NaN for a zone without a valid cell) at ~1/6 of the traffic.
 * majority_dev holds n_zones + 1 doubles: the LAST one is the number of parts whose table overflowed (a zone of more
 * than ~2^27 cells of all-distinct values) -- nonzero means the results are not valid and the caller must use
 * xrs_zonal_majority_*.  n_zones <= xrs_zonal_mode_max_zones(); n < 2^31; `work_dev` holds
 * xrs_zonal_mode_workspace_bytes(n, n_zones, values_f64) bytes.  zone_counts_dev: the valid cells per zone as uint32 if the
 * caller has them (the `count` of xrs_zonal_partials_* for the same rasters and nodata value), else NULL -- they are
 * then counted here with one more pass over the rasters. */
size_t xrs_zonal_mode_workspace_bytes(int64_t n, int n_zones, int values_f64);
int xrs_zonal_mode_max_zones(void);
int xrs_zonal_mode_f32(const int32_t *zone_idx_dev, const float *values_dev, int64_t n, int n_zones, float nodata,
                       int has_nodata, const uint32_t *zone_counts_dev, void *work_dev, size_t work_bytes, double *majority_dev,
                       void *stream);
int xrs_zonal_mode_f64(const int32_t *zone_idx_dev, const double *values_dev, int64_t n, int n_zones, double nodata,
                       int has_nodata, const uint32_t *zone_counts_dev, void *work_dev, size_t work_bytes, double *majority_dev,
                       void *stream);
/* the valid cells of every zone gathered into one contiguous run, for statistics that are arbitrary host callables
 * (zonal.stats(stats_funcs={name: callable}): _calc_stats, xrspatial/zonal.py:144-163, slices the argsort-ordered
 * values per zone and filters non-finite / nodata cells before calling func).  sorted_values_dev[n] receives the cells
 * ordered by (zone index, value ascending); cells outside [0, n_zones) or with an invalid value come last (as NaN), so
 * zone z's values are the slice [sum(count[:z]), sum(count[:z+1])) with count from xrs_zonal_partials_*.  Workspace:
 * xrs_zonal_majority_workspace_bytes(n, n_zones, values_f64).  n < 2^31 cells per call. */
int xrs_zonal_group_f32(const int32_t *zone_idx_dev, const float *values_dev, int64_t n, int n_zones, float nodata,
                        int has_nodata, void *work_dev, size_t work_bytes, float *sorted_values_dev, void *stream);
int xrs_zonal_group_f64(const int32_t *zone_idx_dev, const double *values_dev, int64_t n, int n_zones, double nodata,
                        int has_nodata, void *work_dev, size_t work_bytes, double *sorted_values_dev, void *stream);

/* return_type='xarray.DataArray' of zonal.stats (xrspatial/zonal.py:313-332): out[s][cell] =
 * table[s][zone_idx[cell]] (row-major n_stats x n_zones float64 table), NaN where the cell has no zone. */
int xrs_zonal_backproject_f64(const int32_t *zone_idx_dev, int64_t n, const double *table_dev, int n_stats,
                              int n_zones, double *out_dev, void *stream);

/* zonal.trim / zonal.crop (xrspatial/zonal.py:1651-1731 `_trim`, :1845-1940 `_crop`): bounding box of the cells
 * that equal one of `values` (host array, at most 16; invert = 0: crop's zone ids) or equal none of them
 * (invert = 1: trim's nodata values).  Cells are compared as float64, NaN equals nothing (`e == val` upstream).
 * box4_dev = { top, bottom, left, right }; { rows, -1, cols, -1 } when no cell qualifies.  `dtype`: XRS_DT_*. */
int xrs_match_bbox(const void *data_dev, int dtype, int64_t rows, int64_t cols, int64_t ld, const double *values,
                   int n_values, int invert, int *box4_dev, void *stream);

/* multispectral.true_color (xrspatial/multispectral.py:1334-1495).
 *   xrs_nan_minmax_f32: minmax_dev[0..1] = np.nanmin / np.nanmax of a float32 plane (NaN, NaN if it holds no number);
 *   xrs_true_color_u8:  rgba[i] = { stretch(red), stretch(green), stretch(blue), alpha } with
 *       stretch(v) = uint8( float32( 255 / (1 + exp(c * (th - (v - min) / (max - min)))) ) )   (0 where max == min or NaN)
 *       alpha      = 0 where the red band, in its OWN dtype (`red_raw_dev`, XRS_DT_*), is NaN or <= nodata, else 255;
 *     minmax6_dev = { rmin, rmax, gmin, gmax, bmin, bmax } as produced by xrs_nan_minmax_f32. */
int xrs_nan_minmax_f32(const float *in_dev, int64_t n, float *minmax_dev, void *stream);
int xrs_true_color_u8(const float *red_dev, const float *green_dev, const float *blue_dev, const void *red_raw_dev,
                      int red_raw_dtype, int64_t n, const float *minmax6_dev, double nodata, double c, double th,
                      unsigned char *rgba_dev, void *stream);

/* ----------------------------------------------------- multi-GPU (RCCL, xGMI)
 * One process per GPU.  Rank 0 creates a 128-byte id and ships it to the other
 * ranks by any out-of-band means; every rank then calls xrs_comm_init_rank.
 * halo exchange: the raster is sharded on the row axis; `shard_dev` points at
 * the first OWNED row of this rank's shard, which must have `halo` spare rows
 * above and below it.  One grouped ncclSend/ncclRecv pair per neighbour fills
 * them (rank r-1 above, r+1 below); outer ranks' outer halos are left untouched.
 * zonal reduce: element-wise all-reduce of the partial arrays (sum for
 * count/sum/sumsq, min, max) so every rank holds the global partials; minmax_f64 says
 * whether min_dev / max_dev hold float64 (the *_f64 partials) or float32. */
int xrs_comm_unique_id(void *id128);
int xrs_comm_init_rank(void **comm, const void *id128, int nranks, int rank);
int xrs_comm_destroy(void *comm);
/* {RCCL version code, ranks in the communicator, this rank, HIP device} as RCCL reports them (diagnostics:
 * `bench.py --dry-rccl`; the reference has no counterpart -- dask's scheduler dashboard plays that role) */
int xrs_comm_info(void *comm, int *info4);
int xrs_halo_exchange_f32(void *comm, float *shard_dev, int64_t rows, int64_t cols, int64_t ld,
                          int halo, void *stream);
/* single-GPU loop-back check of the RCCL send/recv plumbing (test support, not on the data path) */
int xrs_comm_selftest_f32(void *comm, const float *src_dev, float *dst_dev, int64_t count, void *stream);
int xrs_zonal_allreduce(void *comm, uint64_t *count_dev, double *sum_dev, double *sumsq_dev,
                        void *min_dev, void *max_dev, int minmax_f64, int n_zones, void *stream);
/* plain typed all-reduce of a device buffer, in place; op: 0 = sum, 1 = min, 2 = max.  Control-plane values of the
 * sharded operators (what dask's scheduler moves for the reference: the zone-id range and the presence map of
 * zonal.stats, zonal.py:181-277; the global moments of hotspots, focal.py:940-984) and the benchmark's barrier. */
int xrs_allreduce_f64(void *comm, double *buf_dev, int64_t count, int op, void *stream);
int xrs_allreduce_u8(void *comm, uint8_t *buf_dev, int64_t count, int op, void *stream);
int xrs_allreduce_u64(void *comm, uint64_t *buf_dev, int64_t count, int op, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* XRS_HIP_H */
