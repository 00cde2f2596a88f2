#!/bin/bash
# Experimental builds side by side: bench.py once per library given on the command line.
for so in "$@"; do
  NSDECODE_LIB=$PWD/$so timeout -k 10 200 python bench.py --cpu-seconds 0.5 --also U > gpurun_out/b_$(basename $so).log 2>&1 || { tail -3 gpurun_out/b_$(basename $so).log; exit 1; }
  python - "$so" gpurun_out/b_$(basename $so).log <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], "I ms/decode %.3f launch %.2f us | U ms/decode %.3f launch %.2f us" % (
    d["ms_per_step"], d["roofline"]["avg_launch_us"], d["variant_U"]["ms_per_decode"],
    d["variant_U"]["roofline"].get("avg_launch_us", float("nan"))))
PY
done
