#!/bin/bash
# Same-box A/B of two libraries on clean and nodata rasters: bash tools/ab_nan.sh <out dir> lib1.so lib2.so ...
OUT=$1; shift
mkdir -p $OUT
CASES=hillshade,slope,terrain_fused4,pass_hill_focal5,pass_hill_slope_focal5,pass_all4_focal5,pass_aspect_focal5,pass_hill_focal3,focal5_mean,focal3_mean,focal_mean3x3_f64,copy_kernel
for rep in 1 2; do
for lib in "$@"; do
  echo "--- $lib (round $rep)"
  XRS_LIB=$PWD/xrspatial_amd/$lib timeout 300 python tools/kbench.py --reps 20 --only $CASES --fast-inputs 2>&1 | grep -v "^inputs\|^device" 
done
done 2>&1 | tee $OUT/ab_kbench.log
for lib in "$@"; do
  echo "--- $lib nan_probe"
  XRS_LIB=$PWD/xrspatial_amd/$lib NAN_PROBE_CASES=${NAN_PROBE_CASES:-hillshade,focal5_mean,fused,focal.mean,focal25_mean,focal25_stats7,box5_mean} timeout 600 python tools/nan_probe.py
done 2>&1 | tee $OUT/ab_nan_probe.log
