#!/bin/bash
# Round evidence on the GPU box: default bench line, rocprofv3 kernel stats of the same command, and the PMC
# passes (one counter group per run) for nsd_step.  Outputs under gpurun_out/<tag>/.
set -e -o pipefail
tag=${1:-final}
root=$PWD
out=$root/gpurun_out/$tag
rm -rf "$out"; mkdir -p "$out"
timeout -k 10 400 python bench.py > "$out/bench.log" 2>&1
tail -1 "$out/bench.log" > "$out/bench.json"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o s -- python3 "$root/bench.py" --no-cpu > "$out/stats.log" 2>&1
rm -f "$out"/stats/*kernel_trace.csv
B="python3 $root/bench.py --large-utts 0 --no-cpu --no-events --steps 4 --warmup 1 --also="
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- $B > "$out/pmc_fetch.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- $B > "$out/pmc_write.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/pmc_l2" -- $B > "$out/pmc_l2.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d "$out/pmc_sq" -- $B > "$out/pmc_sq.log" 2>&1
cd "$root"
python3 scripts/collect_traffic.py "$out/pmc_fetch" "$out/pmc_write" "$out/pmc_l2" "$out/pmc_traffic.json" "configs[1]: 4096 utterances, beam 16, variant I, default bench.py flags"
python3 - "$out" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
f = glob.glob(f"{out}/pmc_sq/*/*counter_collection.csv")[0]
agg, n = collections.defaultdict(float), collections.Counter()
for r in csv.DictReader(open(f)):
    if "nsd_step" in r["Kernel_Name"]:
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
per = {k: agg[k] / n[k] for k in agg}
per["launches"] = max(n.values())
if per.get("SQ_WAVES"):
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
        per[k + "_per_wave"] = per[k] / per["SQ_WAVES"]
    per["wait_any_share_of_wave_cycles"] = per["SQ_WAIT_ANY"] / per["SQ_WAVE_CYCLES"]
json.dump(per, open(f"{out}/pmc_sq.json", "w"), indent=1)
print(per)
PY
find "$out" -name "*.csv" -size +2M -delete
du -sh "$out"
