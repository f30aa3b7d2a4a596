#!/bin/bash
# One GPU-box session: test-suite with the parity report, large-window focal parity + A/B timing, bench line, PMC
# counters, rocprofv3 kernel stats, per-kernel table.  Every step has its own timeout and log; nothing stops the rest.
#   gpurun --timeout 1800 -- 'bash tools/gpu_round.sh r02a [steps...]'
TAG=${1:-r02x}; shift
STEPS=${@:-"tests focal bench pmc stats kbench"}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ROOT=$(pwd)
echo "build id: $(python -c 'from xrspatial_amd import _lib; print(_lib.build_id())')" | tee $OUT/build_id.txt
for s in $STEPS; do
  t0=$(date +%s)
  case $s in
    tests)
      XRS_PARITY_REPORT=$ROOT/$OUT/parity.json timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider \
        --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -40 $OUT/pytest_gpu.log ;;
    focal)
      timeout 900 python tests/focal_large_check.py --out $OUT/focal_large.json > $OUT/focal_large.log 2>&1
      echo "rc=$?" >> $OUT/focal_large.log; grep -v "^ok " $OUT/focal_large.log | tail -60 ;;
    bench)
      timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json; tail -5 $OUT/bench.err ;;
    pmc)
      timeout 900 python tools/pmc_profile.py copy_kernel,geodesic_slope,geodesic_aspect,pass_hill_focal5,pass_hill_slope_focal5,focal5_mean,hillshade,slope,aspect,focal25_mean,focal25_stats7,focal25_meanvarstd,focal25_minmaxrange,focal5_stats7,focal7_stats7,box25_stats7,box25_meanvarstd,box25_mean,annulus21_stats7,annulus21_mean,annulus25_mean,convolve21_annulus,terrain_hill_aspect_curv,zonal_1000,zonal_1000_scattered,zonal_5000 \
        --out $OUT/pmc.json --traffic $OUT/pmc_traffic.json > $OUT/pmc.log 2>&1; echo "pmc rc=$?"; tail -3 $OUT/pmc.log ;;
    stats)
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stats_$TAG -o s -- \
        python $ROOT/bench.py --steps 50 --warmup 10 --no-extras --no-cpu-baseline > $ROOT/$OUT/bench_under_rocprof.json 2> $ROOT/$OUT/bench_under_rocprof.err)
      find /tmp/stats_$TAG -name "*kernel_stats.csv" -exec cp {} $OUT/bench_kernel_stats.csv \; ; cat $OUT/bench_under_rocprof.json | head -c 600; echo; head -5 $OUT/bench_kernel_stats.csv ;;
    kbench)
      timeout 900 python tools/kbench.py --reps 10 --json $OUT/kbench.json > $OUT/kbench.log 2>&1; echo "kbench rc=$?"; tail -70 $OUT/kbench.log ;;
    focal_ab)
      for lib in libxrs_hip.so libxrs_hip_u10.so libxrs_hip_u25.so; do
        echo "--- $lib"; XRS_LIB=$ROOT/xrspatial_amd/$lib timeout 600 python tests/focal_large_check.py --skip-parity --out $OUT/focal_timing_$lib.json 2>&1 | grep -v "^ok " | grep "r=12\|r=6\|copy"
      done ;;
    quicktests)
      timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "focal or fused or pass or kxk or circular or wide or window or full_size or multispectral or flat or large_mask" > $OUT/pytest_quick.log 2>&1
      echo "rc=$?" >> $OUT/pytest_quick.log; tail -15 $OUT/pytest_quick.log ;;
    kb_small)
      timeout 600 python tools/kbench.py --reps 10 --only copy_kernel,stream_1r2w,stream_1r3w,stream_1r7w,hillshade,slope,aspect,curvature,terrain_fused4,pass_aspect_focal5,pass_all4_focal5,pass_hill_focal5,pass_hill_slope_focal5,pass_curv_hill_focal5,pass_hill_focal3,focal5_mean,focal3_mean,focal25_mean,focal25_stats7,focal13_mean,focal13_stats7,box11_mean,box11_stats7 --fast-inputs > $OUT/kb_small.log 2>&1; tail -20 $OUT/kb_small.log ;;
    nan)
      timeout 600 python tools/nan_probe.py > $OUT/nan_probe.log 2>&1; echo "nan rc=$?"; grep -E "focal25|fused|hillshade" $OUT/nan_probe.log | tail -30 ;;
    n1)
      # the one-GPU references of the strong-scaling workloads (bench.py quotes speedup_vs_n1 against them): -> profiles/n1_strong.json
      timeout 600 python bench.py --workload s64 --steps 10 --warmup 3 --write-n1 $OUT/n1_strong.json > $OUT/bench_s64.json 2> $OUT/bench_s64.err; cat $OUT/bench_s64.json | head -c 400; echo
      timeout 600 python bench.py --workload zonal32k --steps 10 --warmup 3 --write-n1 $OUT/n1_strong.json > $OUT/bench_zonal32k.json 2> $OUT/bench_zonal32k.err; cat $OUT/bench_zonal32k.json | head -c 400; echo
      cat $OUT/n1_strong.json ;;
    fuzz)
      for seed in 51 52; do timeout 600 python tests/fuzz_parity.py --seed $seed --cases 1500 > $OUT/fuzz_s$seed.log 2>&1; tail -3 $OUT/fuzz_s$seed.log; done
      timeout 600 python tests/fuzz_parity.py --seed 53 --cases 40 --big > $OUT/fuzz_big.log 2>&1; tail -3 $OUT/fuzz_big.log
      # large windows on rasters with nodata regions, cliffs, lakes, spikes, +-inf (the large-window walkers' whole cascade)
      for seed in 61 62 63 64; do timeout 600 python tests/fuzz_parity.py --windows --seed $seed --cases 500 > $OUT/fuzz_win_s$seed.log 2>&1; tail -2 $OUT/fuzz_win_s$seed.log; done
      # ... and every operator on the same kind of raster
      for seed in 71 72; do timeout 600 python tests/fuzz_parity.py --structured --seed $seed --cases 1000 > $OUT/fuzz_struct_s$seed.log 2>&1; tail -1 $OUT/fuzz_struct_s$seed.log; done ;;
    momnan)
      # per-kernel durations of the large-window moments / extrema kernels on the benchmark DEM with 0.1 % nodata
      (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/momnan_$TAG -o s -- python $ROOT/tools/mom_nan_prof.py > /dev/null 2>&1)
      find /tmp/momnan_$TAG -name "*kernel_stats.csv" -exec cp {} $OUT/mom_nan_kernel_stats.csv \; ; cut -c1-180 $OUT/mom_nan_kernel_stats.csv | head -8 ;;
    region)
      # the large-window kernels on a raster whose first third is ONE nodata region (tools/mom_region_prof.py), per kernel
      for r in rows third; do
        (cd /tmp && REGION=$r timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/region_${TAG}_$r -o s -- python $ROOT/tools/mom_region_prof.py > /dev/null 2>&1)
        find /tmp/region_${TAG}_$r -name "*kernel_stats.csv" -exec cp {} $OUT/region_${r}_kernel_stats.csv \; ; echo "REGION=$r"; cut -d, -f1-4 $OUT/region_${r}_kernel_stats.csv | cut -c1-160 | head -6
      done ;;
    majority)
      timeout 300 python tools/majority_probe.py 32768 > $OUT/majority_probe.log 2>&1; cat $OUT/majority_probe.log
      (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/maj_$TAG -o s -- python $ROOT/tools/majority_probe.py 32768 > /dev/null 2>&1)
      find /tmp/maj_$TAG -name "*kernel_stats.csv" -exec cp {} $OUT/majority_kernel_stats.csv \; ; cut -d, -f1-4 $OUT/majority_kernel_stats.csv | cut -c1-200 | head -24 ;;
    s64bench)
      timeout 600 python bench.py --workload s64 --steps 10 --warmup 3 > $OUT/bench_s64.json 2> $OUT/bench_s64.err; cat $OUT/bench_s64.json ;;
    *) echo "unknown step $s" ;;
  esac
  echo "== step $s took $(( $(date +%s) - t0 )) s"
done
