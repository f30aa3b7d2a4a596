#!/bin/bash
# Scalar-register pressure check for the large-window moments kernel (mom_impl.h): compiles the radius-12 circle
# instantiation alone and counts the v_readlane_b32 (SGPRs parked in VGPR lanes) and VALU instructions of the interior
# walker's round loop.  8 / 610 at the time of writing; control-flow changes AFTER the loop have pushed it to 48-56 and
# cost 5 % of the kernel.   usage: tools/readlanes.sh ["extra compiler flags"]
set -e
cd "$(dirname "$0")/../xrspatial_amd/csrc"
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -Xclang -target-feature -Xclang -packed-fp32-ops \
    -DXRS_MOM_PROBE $1 -c kxk_mom_circle.hip -o $T/probe.o -save-temps=obj 2>&1 | grep -v "not a recognized feature" || true
S=$T/kxk_mom_circle-hip-amdgcn-amd-amdhsa-gfx950.s
K=_ZN12_GLOBAL__N_116focal_mom_kernelILi12EN3xrs11CircleShapeELi14EEEvNS_7MomArgsE
awk -v k="^$K:" '$0 ~ k {f=1} f{print} /^\.Lfunc_end/{if(f) exit}' $S > $T/kernel.s
grep -E "^\s+\.(name|vgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):" $S | paste - - - - - | grep "$K" | sed -E 's/\s+/ /g'
start=$(grep -n "Inner Loop Header: Depth=1" $T/kernel.s | head -1 | cut -d: -f1)
end=$((start + 1080))
echo "round loop (lines $start..$end): v_readlane $(sed -n ${start},${end}p $T/kernel.s | grep -c v_readlane)," \
     "VALU $(sed -n ${start},${end}p $T/kernel.s | grep -cE '^\s*v_'), scratch $(sed -n ${start},${end}p $T/kernel.s | grep -c scratch_)"
rm -rf $T
