"""Register / scratch report for every kernel of libxrs_hip.so (no GPU needed: hipcc -S for gfx950).

A kernel that starts spilling after an innocent-looking change loses half its speed without failing any test (round 1:
the 3x3 focal mean went from 0.42 to 0.96 ms when a NaN-aware body was inlined next to the fast one).  This prints
every kernel that uses scratch memory, with its VGPR count and the number of spilled registers.

    python tools/spill_scan.py [--all] [file.hip ...]
"""
import argparse
import concurrent.futures
import glob
import os
import re
import subprocess
import sys
import tempfile

CSRC = os.environ.get("XRS_CSRC") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "xrspatial_amd", "csrc")   # XRS_CSRC: another checkout (git archive <commit> | tar -x)


def scan(path):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-S", "--cuda-device-only",
                        "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",       # (the Makefile's NOPK)
                        "-o", out, path], check=True, cwd=CSRC)
        text = open(out).read()
    rows = []
    for m in re.finditer(r"- \.agpr_count:.*?\n((?:    .*\n)+)", text):
        blk = m.group(0)
        get = lambda key: re.search(r"\.%s:\s+(\S+)" % key, blk).group(1)      # noqa: E731
        rows.append((os.path.basename(path), get("name"), int(get("vgpr_count")), int(get("private_segment_fixed_size")),
                     int(get("vgpr_spill_count"))))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--all", action="store_true", help="list every kernel, not only the ones with scratch")
    ap.add_argument("files", nargs="*")
    args = ap.parse_args()
    files = args.files or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    with concurrent.futures.ThreadPoolExecutor(8) as pool:
        results = [r for rows in pool.map(scan, files) for r in rows]
    print(f"{'file':22s} {'vgpr':>5s} {'scratch B':>9s} {'spilled':>7s}  kernel")
    n = 0
    for f, name, vgpr, scratch, spill in results:
        if args.all or scratch:
            n += 1
            print(f"{f:22s} {vgpr:5d} {scratch:9d} {spill:7d}  {name}")
    print(f"{len(results)} kernels, {sum(1 for r in results if r[3])} with scratch memory")
    return 0


if __name__ == "__main__":
    sys.exit(main())
