"""Hot-loop census of one kernel (no GPU needed): compiles a translation unit of xrspatial_amd/csrc for gfx950, finds the
smallest loop of the named kernel that holds an LDS-DMA (`global_load_lds`) and a round's worth of vector arithmetic, and
counts its vector / scalar / LDS / scratch instructions.  A spill INSIDE that loop drains the DMA ring at every reload; a
spill in the cold fall-back paths behind it costs nothing.

    python tools/loopscan.py kxk_wide_ann12.hip 'AnnulusShapeILi6EEELi0' ["extra compiler flags"]
"""
import os
import re
import subprocess
import sys
import tempfile

CSRC = os.environ.get("XRS_CSRC") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "xrspatial_amd", "csrc")   # XRS_CSRC: another checkout


def main():
    tu, want = sys.argv[1], sys.argv[2]
    extra = sys.argv[3].split() if len(sys.argv) > 3 else []
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-S", "--cuda-device-only",
                        "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", *extra, "-o", out, tu], check=True, cwd=CSRC)
        text = open(out).read()
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?.*\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", text):
        meta[m.group(1)] = (int(m.group(2)), int(m.group(3)))
    for name in sorted(meta):
        if want not in name:
            continue
        m = re.search(r"^" + re.escape(name) + r":[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M)
        lines = m.group(1).split("\n")
        labels = {mm.group(1): i for i, l in enumerate(lines) for mm in [re.match(r"^(\.LBB\d+_\d+):", l)] if mm}
        best = None
        for lab, i in labels.items():
            ends = [j for j, l in enumerate(lines) if j > i and re.search(r"s_cbranch\w+\s+" + re.escape(lab) + r"\s*$", l)]
            if not ends:
                continue
            body = lines[i:ends[-1] + 1]
            if any("global_load_lds" in l for l in body) and sum(1 for l in body if re.match(r"^\s+v_", l)) > 200 \
                    and (best is None or len(body) < len(best)):
                best = body
        c = (lambda pat: sum(1 for l in best if re.search(pat, l))) if best else (lambda pat: -1)
        pats = dict(valu=r"^\s+v_", salu=r"^\s+s_", lds=r"^\s+ds_", scratch=r"scratch_", readlane=r"v_readlane",
                    dma=r"global_load_lds", stores=r"global_store")
        print(name[:110])
        print(f"   vgpr {meta[name][0]} spilled {meta[name][1]} | loop: " + " ".join(f"{k} {c(v)}" for k, v in pats.items()))

if __name__ == "__main__":
    main()
