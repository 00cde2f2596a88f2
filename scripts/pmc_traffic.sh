#!/bin/bash
# HBM-side traffic of nsd_step for any bench.py workload: three rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE,
# TCC hit / miss), summarised by scripts/collect_traffic.py.   usage: pmc_traffic.sh <tag> "<workload text>" [bench.py flags...]
set -e -o pipefail
tag=$1; text=$2; shift 2
root=$PWD
out=$root/gpurun_out/$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --large-utts 0 --no-cpu --no-events --steps 3 --warmup 1 --also= $*"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- $B > "$out/pmc_fetch.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- $B > "$out/pmc_write.log" 2>&1
timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/pmc_l2" -- $B > "$out/pmc_l2.log" 2>&1
cd "$root"
python3 scripts/collect_traffic.py "$out/pmc_fetch" "$out/pmc_write" "$out/pmc_l2" "$out/pmc_traffic.json" "$text"
find "$out" -name "*.csv" -delete
