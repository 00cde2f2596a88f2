#!/bin/bash
# nsd_step launch time against lanes per utterance at several batch sizes (variant I, beam 16).
out=gpurun_out/$1; mkdir -p $out
for cfg in "4096 32" "4096 64" "16384 32" "16384 64" "32768 32" "32768 64" "131072 16" "131072 32" "131072 64"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --large-utts 0 --no-cpu --utts $1 --lanes $2 --steps 4 --warmup 1 --also '' --timed-decodes 2 > $out/l_$1_$2.json 2> $out/l_$1_$2.err || { echo fail $cfg; exit 1; }
  python - $out/l_$1_$2.json $1 $2 <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("utts", sys.argv[2], "lanes", sys.argv[3], "launch us %.1f frac %.3f Mchars/s %.1f" % (r["avg_launch_us"], r["frac"], d["value"] / 1e6))
PY
done
