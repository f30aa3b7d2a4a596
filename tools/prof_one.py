"""Run ONE kernel a few times on a 16384^2 raster (for rocprofv3 --pmc / --kernel-trace runs).
    python tools/prof_one.py <case> [reps]      cases: see tools/kbench.py"""
import os
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
case = sys.argv[1]
reps = sys.argv[2] if len(sys.argv) > 2 else "3"
sys.exit(subprocess.call([sys.executable, os.path.join(here, "kbench.py"), "--size", "16384", "--reps", reps,
                          "--only", case]))
