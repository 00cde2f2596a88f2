#!/bin/bash
# Experimental builds side by side on one box: bench.py (default workload, no CPU baseline) once per library given
# on the command line, twice each, alternating, so that run-to-run drift shows.  An argument is a library path,
# optionally followed by :MODE (exported as NSD_STEP_MODE: which nsd_step instantiations the library may use).
mkdir -p gpurun_out
for rep in 1 2; do
for arg in "$@"; do
  so=${arg%%:*}; mode=${arg#*:}; [ "$mode" = "$arg" ] && mode=9
  tag=$(basename $so)_m$mode
  NSD_STEP_MODE=$mode NSDECODE_LIB=$PWD/$so timeout -k 10 200 python bench.py --large-utts 0 --no-cpu --also U > gpurun_out/b_$tag.log 2>gpurun_out/b_$tag.err || { tail -3 gpurun_out/b_$tag.err; exit 1; }
  python - "$arg" gpurun_out/b_$tag.log <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], "I ms/decode %.3f launch %.2f us busiest %.1f | U ms/decode %.3f launch %.2f us" % (
    d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["busiest_launch"]["us"], d["variant_U"]["ms_per_decode"],
    d["variant_U"]["roofline"].get("avg_launch_us", float("nan"))))
PY
done
done
