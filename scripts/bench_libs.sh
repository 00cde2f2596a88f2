#!/bin/bash
# Experimental builds side by side on one box: bench.py (default workload, no CPU baseline) once per library given
# on the command line, twice each, alternating, so that run-to-run drift shows.
mkdir -p gpurun_out
for rep in 1 2; do
for so in "$@"; do
  NSDECODE_LIB=$PWD/$so timeout -k 10 200 python bench.py --no-cpu --also U > gpurun_out/b_$(basename $so).log 2>gpurun_out/b_$(basename $so).err || { tail -3 gpurun_out/b_$(basename $so).err; exit 1; }
  python - "$so" gpurun_out/b_$(basename $so).log <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], "I ms/decode %.3f launch %.2f us busiest %.1f | U ms/decode %.3f launch %.2f us" % (
    d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["busiest_launch"]["us"], d["variant_U"]["ms_per_decode"],
    d["variant_U"]["roofline"].get("avg_launch_us", float("nan"))))
PY
done
done
