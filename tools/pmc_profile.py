"""Collect rocprofv3 PMC counters for kbench cases, in separate passes (SQ: 8 slots, TCC: 4 with FETCH_SIZE=3 /
WRITE_SIZE=2), and write a per-kernel summary.  Runs on the GPU box:

    python tools/pmc_profile.py <case>[,<case>...] [--out gpurun_out/pmc_<tag>.json] [--traffic profiles/pmc_traffic.json]

All cases run in ONE kbench process per counter pass (6 rocprofv3 runs in total, whatever the number of cases).
Counters are collected with --kernel-trace only (never with other trace domains).  HBM bytes follow
MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide
coalesced reads, so `hbm_read_bytes` = 2 * FETCH_SIZE * 1024 (calibration: include the `copy_kernel` case, whose 1 GiB
read must come out as ~1.07e9 bytes).

Every summary carries the library's build id (xrs_build_id: a hash of the sources it was built from) and the full kernel
symbols; `--traffic` (re)writes the small table bench.py reads for `roofline.traffic`, which bench.py reports only when
the build id of the library it has loaded equals the one recorded here.
"""
import argparse
import csv
import glob
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
PASSES = [
    ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM",
     "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE"],
    ["SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SALU",
     "SQ_INST_CYCLES_VMEM_RD", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"],
    ["FETCH_SIZE", "TCC_MISS"],
    ["WRITE_SIZE", "TCC_HIT", "TCC_REQ"],
    ["TCP_TOTAL_CACHE_ACCESSES", "TCP_TCC_READ_REQ", "TCP_PENDING_STALL_CYCLES", "TCP_TCR_TCP_STALL_CYCLES"],
    ["SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64",
     "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32", "SQ_THREAD_CYCLES_VALU"],
]


def short_name(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    k = k[:k.rindex(">(") + 1] if ">(" in k else k.split("(")[0]                  # drop the argument list
    return re.sub(r"\s+", " ", k).strip()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", help="comma-separated kbench cases")
    ap.add_argument("--out", default="")
    ap.add_argument("--traffic", default="", help="also (re)write this pmc_traffic.json for bench.py")
    ap.add_argument("--reps", default="3")
    ap.add_argument("--size", default="16384")
    ap.add_argument("--env", nargs="*", default=[])
    ap.add_argument("--passes", default="")
    args = ap.parse_args()
    env = dict(os.environ, TMPDIR="/tmp")
    for kv in args.env:
        k, v = kv.split("=", 1)
        env[k] = v
    tag = re.sub(r"[^A-Za-z0-9_]+", "_", args.cases)[:60]
    outdir = f"/tmp/pmc_{tag}"
    per_kernel = {}
    which = [int(i) for i in args.passes.split(",")] if args.passes else range(len(PASSES))
    for pi in which:
        d = os.path.join(outdir, f"pass{pi}")
        cmd = ["rocprofv3", "--pmc", *PASSES[pi], "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
               sys.executable, os.path.join(HERE, "kbench.py"), "--size", args.size, "--reps", args.reps,
               "--only", args.cases, "--fast-inputs"]
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
        if r.returncode != 0:
            print(f"pass {pi} failed:\n{r.stdout[-2000:]}\n{r.stderr[-2000:]}")
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = short_name(row.get("Kernel_Name", "?"))
                    cn, cv = row.get("Counter_Name"), float(row.get("Counter_Value", 0))
                    ent = per_kernel.setdefault(k, {})
                    tot, n = ent.get(cn, (0.0, 0))
                    ent[cn] = (tot + cv, n + 1)
    from xrspatial_amd import _lib
    summary = {"_build_id": _lib.build_id(), "_cases": args.cases, "_size": int(args.size),
               "_method": "rocprofv3 --pmc <one counter group per run> --kernel-trace; per-dispatch averages; "
                          "hbm_read_bytes = 2 * FETCH_SIZE * 1024 (gfx950 half-count correction, MI355X_MICROARCH.md), "
                          "hbm_write_bytes = WRITE_SIZE * 1024"}
    for k, ent in per_kernel.items():
        if "kernel" not in k.lower() and "copy" not in k.lower():
            continue
        s = {cn: tot / n for cn, (tot, n) in ent.items()}      # per-dispatch averages
        if "FETCH_SIZE" in s:
            s["hbm_read_bytes"] = 2 * s["FETCH_SIZE"] * 1024
        if "WRITE_SIZE" in s:
            s["hbm_write_bytes"] = s["WRITE_SIZE"] * 1024
        if "hbm_read_bytes" in s and "hbm_write_bytes" in s:
            s["hbm_bytes"] = s["hbm_read_bytes"] + s["hbm_write_bytes"]
        if "SQ_ACTIVE_INST_VALU" in s and s.get("SQ_WAVE_CYCLES"):
            s["valu_active_frac_of_wave_cycles_x4"] = 4 * s["SQ_ACTIVE_INST_VALU"] / s["SQ_WAVE_CYCLES"]
        summary[k] = s
    text = json.dumps(summary, indent=1, sort_keys=True)
    print(text)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as fh:
            fh.write(text)
    if args.traffic:
        table = {"_build_id": summary["_build_id"], "_size": summary["_size"], "_source": args.out,
                 "_note": "HBM bytes per launch = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (rocprofv3 --pmc, separate passes; gfx950 "
                          "FETCH_SIZE half-count correction per MI355X_MICROARCH.md); keyed by the full kernel symbol; valid "
                          "only for the library build named in _build_id (bench.py checks xrs_build_id)",
                 "kernels": {k: {"hbm_bytes": v["hbm_bytes"], "hbm_read_bytes": v["hbm_read_bytes"],
                                 "hbm_write_bytes": v["hbm_write_bytes"]}
                             for k, v in summary.items() if isinstance(v, dict) and "hbm_bytes" in v}}
        os.makedirs(os.path.dirname(os.path.abspath(args.traffic)), exist_ok=True)
        with open(args.traffic, "w") as fh:
            json.dump(table, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
