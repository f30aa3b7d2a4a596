#!/bin/bash
# Scalar-register pressure check for the large-window moments kernel (mom_impl.h): compiles the radius-12 circle
# instantiation alone and counts the v_readlane_b32 (SGPRs parked in VGPR lanes) and VALU instructions of the interior
# walker's round loop (5 rows x 2 columns per lane).  At the time of writing: see DESIGN.md; control-flow changes AFTER the loop have pushed it to 48-56 and
# cost 5 % of the kernel.   usage: tools/readlanes.sh ["extra compiler flags"]
set -e
cd "$(dirname "$0")/../xrspatial_amd/csrc"
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -Xclang -target-feature -Xclang -packed-fp32-ops \
    -DXRS_MOM_PROBE $1 -c kxk_mom_circle.hip -o $T/probe.o -save-temps=obj 2>&1 | grep -v "not a recognized feature" || true
S=$T/kxk_mom_circle-hip-amdgcn-amd-amdhsa-gfx950.s
K=_ZN12_GLOBAL__N_116focal_mom_kernelILi12EN3xrs11CircleShapeELi${OMSET:-14}EEEvNS_7MomArgsE
awk -v k="^$K:" '$0 ~ k {f=1} f{print} /^\.Lfunc_end/{if(f) exit}' $S > $T/kernel.s
grep -E "^\s+\.(name|vgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):" $S | paste - - - - - | grep "$K" | sed -E 's/\s+/ /g'
# the interior walker's round loop: the first depth-1 loop that holds LDS-DMA instructions; from its header label to the
# last branch back to that label
python3 - $T/kernel.s <<'PY'
import re, sys
lines = open(sys.argv[1]).read().split("\n")
labels = {}
for i, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
found = []
for lab, i in labels.items():
    ends = [j for j, l in enumerate(lines) if j > i and re.search(r"s_cbranch\w+\s+" + re.escape(lab) + r"\s*$", l)]
    if not ends:
        continue
    body = lines[i:ends[-1] + 1]
    # (every loop with the DMA and a round's worth of arithmetic: the plain walk's round loop, the carrying walk's, and
    # the tile-level loops around them -- the innermost ones are the two shortest)
    if any("global_load_lds" in l for l in body) and sum(1 for l in body if re.match(r"^\s+v_", l)) > 400:
        found.append((len(body), i, ends[-1], body))
if not found:
    print("round loop not found")
for n, i, j, body in sorted(found):
    c = lambda pat: sum(1 for l in body if re.search(pat, l))
    counts = (c("v_readlane"), c("v_writelane"), c("^\\s+v_"), c("^\\s+s_"), c("^\\s+ds_"), c("scratch_"), c("global_load_lds"), c("vmcnt\\(0\\)"))
    print("loop (lines %d..%d): v_readlane %d, v_writelane %d, VALU %d, SALU %d, LDS %d, scratch %d, DMA %d, vmcnt(0) %d" % ((i + 1, j + 1) + counts))
PY
[ -n "$KEEP" ] && cp $T/kernel.s $KEEP; rm -rf $T
