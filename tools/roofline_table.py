"""kbench log -> the markdown roofline table of profiles/rNN/ROOFLINE.md.

    python tools/roofline_table.py profiles/r01/kbench_r01j.log > table.md

Columns: algorithmic bytes per cell (every input plane read once + every output plane written once; tools/kbench.py),
HIP-event median, Mcells/s, algorithmic GB/s, its share of the 8 TB/s HBM3E spec and of the library's own streaming copy
(`copy_kernel`, measured in the same run)."""
import re
import sys


def main(path):
    rows = []
    size = None
    for line in open(path):
        m = re.search(r"raster:\s+(\d+) x (\d+)", line)
        if m:
            size = int(m.group(1)) * int(m.group(2))
        m = re.match(r"^(\w+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)\s+(\d+)\s*$", line)
        if m:
            rows.append((m.group(1), float(m.group(2)), int(m.group(4)), int(m.group(5))))
    copy = next(gbs for name, _, _, gbs in rows if name == "copy_kernel")
    # the plain stream with the same plane mix, measured in the same run: 1 plane read, (B/cell - 4) / 4 planes written
    streams = {8: "copy_kernel", 12: "stream_1r2w", 16: "stream_1r3w", 20: "stream_1r4w", 32: "stream_1r7w"}
    stream_ms = {b: ms for b, nm in streams.items() for name, ms, _, _ in rows if name == nm}
    print("| kernel | B/cell | ms | Mcells/s | GB/s (algorithmic) | % of 8 TB/s | % of measured copy | % of the same-mix stream |")
    print("|---|---|---|---|---|---|---|---|")
    for name, ms, mcells, gbs in rows:
        bpc = round(gbs * 1e9 / (mcells * 1e6)) if mcells else 0
        mix = f"{100 * stream_ms[bpc] / ms:.0f} %" if bpc in stream_ms and ms else "—"
        print(f"| {name} | {bpc} | {ms:.3f} | {mcells} | {gbs} | {100 * gbs / 8000:.0f} % | {100 * gbs / copy:.0f} % | {mix} |")
    print(f"\n(copy_kernel = {copy} GB/s; {size} cells.  Same-mix stream: `stream_1rNw` = one plane read, N written, no arithmetic, "
          "in this run -- it moves by 15 % between boxes and within a process; a kernel that reads its input with a halo cannot "
          "reach 100 %.)" if size else "")


if __name__ == "__main__":
    main(sys.argv[1])
