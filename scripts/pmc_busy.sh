#!/bin/bash
# Where do the SIMDs spend their cycles in nsd_step?  Two SQ counter passes (eight counters each) over a short
# bench run; usage: pmc_busy.sh <tag> [bench.py flags...]   Output: gpurun_out/<tag>/busy.json
set -e -o pipefail
tag=$1; shift
root=$PWD
out=$root/gpurun_out/$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --large-utts 0 --no-cpu --no-events --steps 3 --warmup 1 --also= $*"
timeout -s KILL 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY --output-format csv -d "$out/p1" -- $B > "$out/p1.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAVES --output-format csv -d "$out/p2" -- $B > "$out/p2.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_ANY SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 --output-format csv -d "$out/p3" -- $B > "$out/p3.log" 2>&1
cd "$root"
python3 - "$out" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
per = {}
for p in ("p1", "p2", "p3"):
    f = glob.glob(f"{out}/{p}/*/*counter_collection.csv")[0]
    agg, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        if "nsd_step" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    per.update({k: agg[k] / n[k] for k in agg})
    per["launches"] = max(n.values())
json.dump(per, open(f"{out}/busy.json", "w"), indent=1)
for k, v in sorted(per.items()): print(f"{k:28s} {v:16.1f}")
PY
find "$out" -name "*.csv" -delete
