"""Compare two register tables of tools/spill_scan.py --all (no GPU needed): kernels whose VGPR count moved by a wave-per-SIMD
step or more, and kernels that gained or lost scratch memory.

    python tools/spill_scan.py --all > /tmp/now.txt
    python tools/regdiff.py profiles/r04/spill_scan_r04.txt /tmp/now.txt

Round 4: a one-line numerics fix in terrain_cells.h took three fused raster-pass instantiations from 149 to 206 registers (three
waves per SIMD to two, 0.62 -> 0.69 ms) without failing any test; this comparison is how it was found."""
import sys


def table(path):
    rows = {}
    for line in open(path):
        parts = line.split()
        if len(parts) == 5 and parts[1].isdigit():
            rows[(parts[0], parts[4])] = (int(parts[1]), int(parts[2]), int(parts[3]))
    return rows


def waves(vgpr):          # waves per SIMD a kernel's VGPR count allows on gfx950 (512 registers per lane and SIMD, granule 8)
    return min(8, 512 // max(8, -(-vgpr // 8) * 8))


def main(a_path, b_path):
    a, b = table(a_path), table(b_path)
    n = 0
    for key in sorted(set(a) | set(b)):
        if key not in a or key not in b:
            print(("new     " if key in b else "gone    ") + f"{key[0]:22s} {key[1]}")
            continue
        (va, sa, pa), (vb, sb, pb) = a[key], b[key]
        if waves(va) != waves(vb) or (sa == 0) != (sb == 0) or abs(pb - pa) >= 8:
            n += 1
            print(f"changed {key[0]:22s} vgpr {va} -> {vb} (waves/SIMD {waves(va)} -> {waves(vb)}), scratch {sa} -> {sb} B, "
                  f"spilled {pa} -> {pb}  {key[1]}")
    print(f"{n} kernels changed their occupancy class or scratch use ({len(a)} / {len(b)} kernels in the tables)")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2]))
