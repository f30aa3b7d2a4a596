"""Collect rocprofv3 PMC counters for one kbench case, in separate passes (SQ: 8 slots, TCC: 4 with
FETCH_SIZE=3 / WRITE_SIZE=2), and print/write a per-kernel summary.  Runs on the GPU box:

    python tools/pmc_profile.py <case> [--out gpurun_out/pmc_<case>.json] [--env K=V ...]

Counters are collected with --kernel-trace only (never with other trace domains).  HBM bytes follow
MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of
wide coalesced reads, so `hbm_read_bytes` = 2 * FETCH_SIZE * 1024 (the copy_d2d case of the same run is the
calibration: its 1 GiB read must come out as ~1.07e9 bytes).
"""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("case")
    ap.add_argument("--out", default="")
    ap.add_argument("--reps", default="3")
    ap.add_argument("--env", nargs="*", default=[])
    ap.add_argument("--passes", default="")
    args = ap.parse_args()
    env = dict(os.environ, TMPDIR="/tmp")
    for kv in args.env:
        k, v = kv.split("=", 1)
        env[k] = v
    outdir = f"/tmp/pmc_{args.case}"
    per_kernel = {}
    which = [int(i) for i in args.passes.split(",")] if args.passes else range(len(PASSES))
    for pi in which:
        d = os.path.join(outdir, f"pass{pi}")
        cmd = ["rocprofv3", "--pmc", *PASSES[pi], "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
               sys.executable, os.path.join(HERE, "kbench.py"), "--size", "16384", "--reps", args.reps,
               "--only", args.case, "--fast-inputs"]
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
        if r.returncode != 0:
            print(f"pass {pi} failed:\n{r.stdout[-2000:]}\n{r.stderr[-2000:]}")
            continue
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        for f in files:
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = row.get("Kernel_Name", "?")
                    short = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                    cn, cv = row.get("Counter_Name"), float(row.get("Counter_Value", 0))
                    ent = per_kernel.setdefault(short, {})
                    tot, n = ent.get(cn, (0.0, 0))
                    ent[cn] = (tot + cv, n + 1)
    summary = {}
    for k, ent in per_kernel.items():
        if "kernel" not in k.lower() and "copy" not in k.lower():
            continue
        s = {cn: tot / n for cn, (tot, n) in ent.items()}      # per-dispatch averages
        if "FETCH_SIZE" in s:
            s["hbm_read_bytes(2xFETCH_SIZE KiB, gfx950 correction)"] = 2 * s["FETCH_SIZE"] * 1024
        if "WRITE_SIZE" in s:
            s["hbm_write_bytes(WRITE_SIZE KiB)"] = s["WRITE_SIZE"] * 1024
        summary[k] = s
    text = json.dumps(summary, indent=1, sort_keys=True)
    print(text)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as fh:
            fh.write(text)


if __name__ == "__main__":
    main()
