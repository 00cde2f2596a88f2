#!/bin/bash
# Large-batch comparison: bench.py at 131072 utterances for each "mode:lanes" pair given.  Usage: gpu_big.sh <tag> m:l ...
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
for arg in "$@"; do
  mode=${arg%%:*}; lanes=${arg#*:}
  NSD_STEP_MODE=$mode timeout -k 10 200 python bench.py --large-utts 0 --no-cpu --utts 131072 --steps 5 --warmup 1 --also= --lanes $lanes > $out/big_m${mode}_l$lanes.json 2> $out/big_m${mode}_l$lanes.err || exit 1
  python - $arg $out/big_m${mode}_l$lanes.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2])); b = d["roofline"]
print("mode:lanes", sys.argv[1], "big launch %.1f us frac %.3f  %.1f Mchars/s" % (b["avg_launch_us"], b["frac"], d["value"] / 1e6))
PY
done
