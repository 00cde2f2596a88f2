#!/bin/bash
# Experimental builds side by side at a batch far beyond what the chip holds: bench.py --utts $UTTS once per library.
UTTS=${UTTS:-131072}
for so in "$@"; do
  NSDECODE_LIB=$PWD/$so timeout -k 10 300 python bench.py --utts $UTTS --steps 6 --warmup 2 --no-cpu > gpurun_out/bb_$(basename $so).log 2>&1 || { tail -3 gpurun_out/bb_$(basename $so).log; exit 1; }
  python - "$so" gpurun_out/bb_$(basename $so).log <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(sys.argv[1], "I ms/decode %.3f launch %.2f us frac %.3f" % (d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"]))
PY
done
