"""One rank of bench.py with the C ABI answered by tests/fake_hip.py (oracle arithmetic on the CPU): lets the
`-m "not gpu"` suite run the multi-rank logic of every bench workload -- shard layout, halo exchange, the fused pass with
halo_top / halo_bot, the boundary check, the zonal partial reduce and its count check -- in a gloo group of 2 processes.
RCCL does not exist here, so the ranks take bench.py's --allow-host-halo transport.  Usage: bench_worker.py <bench args>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests import fake_hip  # noqa: E402

fake_hip.install()

import bench  # noqa: E402

if __name__ == "__main__":
    bench.main()
