#!/bin/bash
# The three benchmark shapes only (no tests).  Usage: scripts/gpu_bench3.sh <tag>
out=gpurun_out/$1; mkdir -p $out
timeout -k 10 200 python bench.py --large-utts 0 --no-cpu > $out/default.json 2> $out/default.err || exit 1
timeout -k 10 200 python bench.py --large-utts 0 --open-lexicon --beam 32 --no-cpu --steps 6 --also '' > $out/open32.json 2> $out/open32.err || exit 1
timeout -k 10 200 python bench.py --large-utts 0 --no-cpu --utts 131072 --steps 5 --warmup 1 > $out/big.json 2> $out/big.err || exit 1
python - $out <<'PY'
import json, sys
for f in ("default", "open32", "big"):
    d = json.load(open(f"{sys.argv[1]}/{f}.json")); r = d["roofline"]
    u = d.get("variant_U", {})
    print(f, round(d["value"] / 1e6, 1), "Mchars/s launch us", round(r["avg_launch_us"], 2), "frac", round(r["frac"], 3),
          "busiest us", round(r["busiest_launch"]["us"], 1), "| U launch us", round(u.get("roofline", {}).get("avg_launch_us", 0), 2),
          "U Mchars/s", round(u.get("value", 0) / 1e6, 1))
PY
