"""Development aid: hardware counters of the pipeline kernels (rocprofv3 --pmc, one pass per counter group).

    python tools/pmc_profile.py [--mbytes 128] [--out gpurun_out/pmc] [--kernel k_match_branch]

Prints per-dispatch averages for every kernel whose name contains --kernel (default: all tm kernels) as JSON."""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = [
    ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_BRANCH"],
    ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"],
    ["SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_SALU", "GRBM_GUI_ACTIVE"],
    ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum"],
    ["FETCH_SIZE"],      # HBM traffic, one counter per pass as MI355X_MICROARCH.md prescribes (KB; x2 correction on gfx950)
    ["WRITE_SIZE"],
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mbytes", type=int, default=128)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "pmc"))
    ap.add_argument("--kernel", default="")
    ap.add_argument("--extra", default="", help="extra bench.py flags")
    ap.add_argument("--e2e", action="store_true", help="profile the end-to-end step (device normalizer included) instead of the hot path")
    ap.add_argument("--groups", default="", help="comma separated group numbers (default all)")
    ap.add_argument("--fast", action="store_true", help="drive the kernels with tools/k1_time.py (ctypes, no torch: seconds instead of a minute per pass)")
    ap.add_argument("--lib", default="", help="with --fast: a variant library instead of the product's")
    args = ap.parse_args()
    args.out = os.path.abspath(args.out)
    os.makedirs(args.out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    acc = {}
    want = [int(x) for x in args.groups.split(",")] if args.groups else list(range(len(GROUPS)))
    for gi, grp in enumerate(GROUPS):
        if gi not in want:
            continue
        d = os.path.join(args.out, "g%d" % gi)
        if args.fast:
            work = [sys.executable, os.path.join(ROOT, "tools", "k1_time.py"), "--one", os.path.abspath(args.lib) if args.lib else "", "--mbytes", str(args.mbytes),
                    "--reps", "2"] + (["--e2e"] if args.e2e else []) + args.extra.split()
        else:
            work = [sys.executable, os.path.join(ROOT, "bench.py"), "--mbytes", str(args.mbytes), "--steps", "2", "--warmup", "1",
                    "--verify", "0", "--no-cpu-baseline", "--no-host-to-host"] + ([] if args.e2e else ["--hot-path-only"]) + args.extra.split()
        cmd = ["rocprofv3", "--pmc"] + grp + ["--kernel-trace", "-d", d, "-o", "pmc", "--output-format", "csv", "--"] + work
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            print("group %d failed:\n%s" % (gi, r.stdout.decode(errors="replace")[-2000:]), file=sys.stderr)
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = row["Kernel_Name"].split("(")[0]
                    if args.kernel not in k:
                        continue
                    c = row["Counter_Name"]
                    a = acc.setdefault(k, {}).setdefault(c, [0.0, set()])
                    a[0] += float(row["Counter_Value"])
                    a[1].add(row["Dispatch_Id"])
    out = {k: {c: v[0] / max(1, len(v[1])) for c, v in sorted(cs.items())} for k, cs in acc.items()}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
