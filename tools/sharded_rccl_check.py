"""tests/sharded_worker.py on `world` DISTINCT HIP devices with the RCCL transport (distributed.Comm), stitched and
compared bit for bit with the single-device results -- what test_sharded_api_equals_single_gpu does with host-staged halos
on one shared GPU.  For a node (or a compute-partitioned chip) that exposes more than one device:
    python tools/sharded_rccl_check.py 8
"""
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    tmp = tempfile.mkdtemp(prefix="xrs_rccl_")
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), XRS_DEVICE=str(rank),
                   XRS_TEST_TRANSPORT="rccl", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1",
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "sharded_worker.py"), tmp], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    bad = 0
    for r, p in enumerate(procs):
        out, _ = p.communicate(timeout=600)
        if p.returncode:
            bad += 1
            print(f"rank {r} rc={p.returncode}\n{out.decode()[-2500:]}")
    if bad:
        return 1
    from tests.test_gpu_parity import _sharded_reference_results
    full, zones_full, want, table = _sharded_reference_results()
    parts = [np.load(os.path.join(tmp, f"rank{r}.npz")) for r in range(world)]
    n_diff = 0
    for name, ref in want.items():
        got = np.empty_like(ref)
        for p in parts:
            got[..., int(p["y0"]):int(p["y1"]), :] = p[name]
        diff = int(((got != ref) & ~(np.isnan(got) & np.isnan(ref))).sum())
        ok = diff <= (2 if name == 'hot7' else 0)
        n_diff += 0 if ok else 1
        print(f"{name:16s} cells that differ from the single-device result: {diff} {'OK' if ok else 'FAIL'}")
    for p in parts:
        for col in table.columns:
            np.testing.assert_allclose(p['zonal_' + col].astype(np.float64), np.asarray(table[col], dtype=np.float64), rtol=1e-12)
        np.testing.assert_array_equal(p['zonal_count'].astype(np.int64), np.asarray(table['count']).astype(np.int64))
    print(f"zonal tables of all {world} ranks equal the single-device table (counts exact)")
    print("RCCL sharded check:", "OK" if n_diff == 0 else "FAIL", f"world={world}")
    return 0 if n_diff == 0 else 1


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 2))
