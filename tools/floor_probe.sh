#!/bin/bash
# The 25x25 circular statistics as a FLOOR measurement: the two kernels of focal_stats (moments walker mom_impl.h, extrema
# walker ext_impl.h) built three ways -- as shipped, with the arithmetic stubbed (LDS-DMA ring + the output streams only) and
# with the stores stubbed (DMA ring + arithmetic only) -- and timed on the same box, same run.
#   bash tools/floor_probe.sh build        (here: three libraries into xrspatial_amd/libxrs_hip_floor_*.so)
#   gpurun -- 'bash tools/floor_probe.sh run gpurun_out/floor'
set -e
cd "$(dirname "$0")/.."
CS=xrspatial_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -I$CS/_build -Xclang -target-feature -Xclang -packed-fp32-ops"
if [ "$1" = build ]; then
  mkdir -p /tmp/floor
  for v in noarith:-DXRS_FLOOR_NO_ARITH nostores:-DXRS_FLOOR_NO_STORES; do
    n=${v%%:*}; f=${v#*:}
    # (kxk_wide_circle: the 25x25 mean / convolution walker has the store stub only -- its arithmetic floor is the instruction
    #  census of tools/loopscan.py, profiles/r06/ROOFLINE.md)
    for tu in kxk_mom_circle kxk_ext_circle kxk_wide_circle; do
      (cd $CS && /opt/rocm/bin/hipcc $FLAGS $f -c $tu.hip -o /tmp/floor/${tu}_$n.o 2> >(grep -v "is not a recognized feature" >&2)) &
    done
    wait
    objs=$(ls $CS/_build/*.o | grep -v "/kxk_mom_circle.o\|/kxk_ext_circle.o\|/kxk_wide_circle.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o xrspatial_amd/libxrs_hip_floor_$n.so $objs /tmp/floor/kxk_mom_circle_$n.o /tmp/floor/kxk_ext_circle_$n.o /tmp/floor/kxk_wide_circle_$n.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
  done
  ls -la xrspatial_amd/libxrs_hip_floor_*.so
  exit 0
fi
OUT=${2:-gpurun_out/floor}; mkdir -p $OUT
for rep in 1 2; do
  for lib in libxrs_hip.so libxrs_hip_floor_noarith.so libxrs_hip_floor_nostores.so; do
    echo "--- $lib (round $rep)"
    XRS_LIB=$PWD/xrspatial_amd/$lib timeout 300 python tools/kbench.py --reps 15 --only copy_kernel,stream_1r7w,focal25_stats7,focal25_meanvarstd,focal25_minmaxrange,focal25_mean,convolve25_circle --fast-inputs 2>&1 | grep -v "^inputs\|^device"
  done
done 2>&1 | tee $OUT/floor_probe.log
