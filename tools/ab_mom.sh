for rep in 1 2; do for n in "$@"; do echo "--- $n (round $rep)"; XRS_LIB=$PWD/xrspatial_amd/libxrs_hip_$n.so timeout 300 python tests/focal_large_check.py --skip-parity --out /tmp/x.json 2>&1 | grep "mean+var+std\|all7" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  %-12s %-14s gen2 %.3f ms' % (d['mask'], d['stats'], d['gen2_ms']))"; done; done
