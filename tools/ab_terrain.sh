# same-box A/B of library variants (built by hand into xrspatial_amd/libxrs_hip_<name>.so): bash tools/ab_terrain.sh name1 name2 ...
CASES=${CASES:-hillshade,slope,aspect,curvature,copy_kernel}
for rep in 1 2; do
for n in "$@"; do
  echo "--- $n (round $rep)"
  XRS_LIB=$PWD/xrspatial_amd/libxrs_hip_$n.so timeout 300 python tools/kbench.py --reps 20 --only $CASES --fast-inputs 2>&1 | grep -v "^inputs\|^kernel\|^device"
done
done
