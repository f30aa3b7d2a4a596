#!/bin/bash
# A/B of wide-walker builds on one box: tools/ab_wide.sh "<rows>" lib1 lib2 ...   (libxrs_<name>.so next to the product library)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04w_ab
ROWS=$1; shift
for rep in 1 2; do for v in "$@"; do
  echo "--- lib $v (round $rep)"
  XRS_LIB=$PWD/xrspatial_amd/libxrs_$v.so timeout 300 python tools/kbench.py --only $ROWS --fast-inputs 2>&1 | grep -v "^device\|^inputs\|^kernel "
done; done 2>&1 | tee -a gpurun_out/r04w_ab/ab_wide.log
