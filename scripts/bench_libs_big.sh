#!/bin/bash
# Experimental builds side by side on the 131072-utterance batch (one GPU's share of the 1M-utterance configuration).
mkdir -p gpurun_out
for rep in 1 2; do
for so in "$@"; do
  NSDECODE_LIB=$PWD/$so timeout -k 10 300 python bench.py --large-utts 0 --no-cpu --utts 131072 --steps 4 --warmup 1 --timed-decodes 2 > gpurun_out/bb_$(basename $so).log 2>gpurun_out/bb_$(basename $so).err || { tail -3 gpurun_out/bb_$(basename $so).err; exit 1; }
  python - "$so" gpurun_out/bb_$(basename $so).log <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], "131072 utts: I launch %.1f us frac %.3f | U launch %.1f us" % (
    d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["variant_U"]["roofline"].get("avg_launch_us", float("nan"))))
PY
done
done
