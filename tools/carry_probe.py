"""How many tiles the carrying walk of the moments kernel (mom_impl.h, MomWalk<.., CARRY>) hands on, and what it costs alone.

Needs a probe build beside the default library:  make BUILD=_build_dbg TARGET=../libxrs_hip_dbg.so EXTRA=-DXRS_MOM_CARRY_ONLY
(in that build a NaN tile ends after the carrying walk whether it succeeded or not: tiles it handed on keep whatever the
output buffer held).  The parent runs the child once per library and compares: cells that differ = cells of handed-on tiles.
"""
import os
import subprocess
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = "/tmp/carry_probe"


def child(tag):
    import xrspatial_amd as xs
    from xrspatial_amd import focal
    from xrspatial_amd.convolution import circle_kernel
    from tools.kbench import Timer
    from tests import synth
    t = Timer()
    rng = np.random.default_rng(0)
    os.makedirs(OUT, exist_ok=True)
    for size in (4096, 16384):
        rasters = {"noise": (1000 + rng.random((size, size), dtype=np.float32) * 50), "asv": synth.asv_dem(size, size).copy()}
        for rname, z in rasters.items():
            z[np.random.default_rng(7).random(z.shape) < 0.001] = np.nan
            A = xs.DataArray(xs.DeviceArray.from_numpy(z), dims=["y", "x"], attrs={"res": (1.0, 1.0)})
            for kname, k in (("circle25", circle_kernel(1, 1, 12)), ("box25", np.ones((25, 25)))):
                for sname, st in (("mvs", ['mean', 'var', 'std']), ("stats7", None)):
                    fn = (lambda: focal.focal_stats(A, k, stats_funcs=st)) if st else (lambda: focal.focal_stats(A, k))
                    if size == 4096:
                        # poison whatever buffers the allocator hands out next, then run
                        for _ in range(3):
                            junk = [xs.DeviceArray.from_numpy(np.full((size, size), -7777.0, np.float32)) for _ in range(8)]
                            del junk
                        r = fn().data
                        np.save(f"{OUT}/{tag}_{rname}_{kname}_{sname}.npy", r.get()[:1])
                    else:
                        med, mn = t.time(lambda: (fn(), None)[1], 5, warmup=2)
                        print(f"{tag:5s} {rname:6s} {kname:9s} {sname:7s} {med:8.3f} ms", flush=True)


def main():
    if len(sys.argv) > 1:
        return child(sys.argv[1])
    for tag, lib in (("full", "libxrs_hip.so"), ("carry", "libxrs_hip_dbg.so")):
        env = dict(os.environ, XRS_LIB=os.path.join(ROOT, "xrspatial_amd", lib))
        subprocess.run([sys.executable, os.path.abspath(__file__), tag], env=env, check=True)
    for f in sorted(os.listdir(OUT)):
        if not f.startswith("full_"):
            continue
        a = np.load(f"{OUT}/{f}")[0]
        b = np.load(f"{OUT}/carry_{f[5:]}")[0]
        same = (a == b) | (np.isnan(a) & np.isnan(b))
        rows_bad = (~same).reshape(same.shape[0], -1, 128).any(axis=2)          # 128-column segments (one wave tile wide)
        print(f"{f[5:-4]:28s} cells that differ {100 * (1 - same.mean()):6.2f} %   128-column row segments {100 * rows_bad.mean():6.2f} %")


if __name__ == "__main__":
    main()
