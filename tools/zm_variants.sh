# per-kernel times of the partition-and-count majority (csrc/zonal_mode.hip) on the 32768^2 probe rasters, one raster per rocprofv3 run
#   gpurun -- 'bash tools/zm_variants.sh [lib ...]'     (libs: names under xrspatial_amd/, default libxrs_hip.so)
mkdir -p gpurun_out/zm
for lib in ${@:-libxrs_hip.so}; do
  for c in ${ZM_CASES:-continuous categorical32}; do
    echo "=== $lib $c"
    (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/zm_prof && XRS_LIB=/root/repo/xrspatial_amd/$lib MAJORITY_CASES=$c MAJORITY_SORT=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/zm_prof -o s -- python /root/repo/tools/majority_probe.py 32768 2>/dev/null | grep "^mode")
    f=$(find /tmp/zm_prof -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/zm/${lib%.so}_${c}_kernel_stats.csv
    python - <<PY
import csv
for row in csv.DictReader(open("gpurun_out/zm/${lib%.so}_${c}_kernel_stats.csv")):
    n=row['Name']
    for k in ('zone_count','scatter_zone','part_hist_kernel<unsigned int, false','scatter_part_kernel<unsigned int, false','count_kernel<'):
        if k in n: print("   %-42s %s calls avg %.3f ms" % (k, row['Calls'], float(row['AverageNs'])/1e6))
PY
  done
done
