#!/bin/bash
# Instructions per phase of nsd_step (see scripts/phase_insts.py; build the truncated libraries first, on the CPU box).
# usage: phase_insts.sh <tag> [variant utts beam step]
set -e -o pipefail
tag=$1; shift
root=$PWD
out=$root/gpurun_out/$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_BRANCH SQ_WAVE_CYCLES --output-format csv -d "$out/p" -- python3 "$root/scripts/phase_insts.py" run "$@" > "$out/run.log" 2>&1
cd "$root"
python3 - "$out" <<'PY'
import collections, csv, glob, sys
out = sys.argv[1]
f = glob.glob(f"{out}/p/*/*counter_collection.csv")[0]
per = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if "nsd_step_kernel" in r["Kernel_Name"]:
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(per)[-21:]
names = ["phase 1 (state, records)", "passes (softmax, emit)", "bounds + scoring", "selection: set-up, threshold", "selection: ranking", "selection: holes, ties", "write-back"]
prev = collections.defaultdict(float)
for k in range(7):
    d = per[ids[3 * k]]                      # first of the three launches of this build
    w = d["SQ_WAVES"]
    print(f"{names[k]:28s} " + "  ".join(f"{c[9:]} {(d[c] - prev[c]) / w:7.1f}" for c in
          ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_BRANCH")))
    prev = d
print(f"{'whole kernel':28s} " + "  ".join(f"{c[9:]} {prev[c] / prev['SQ_WAVES']:7.1f}" for c in
      ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_BRANCH")))
PY
find "$out" -name "*.csv" -delete
