#!/bin/bash
# nsd_step launch time with and without the upper trie levels staged in LDS, closed and open lexicon.
out=gpurun_out/$1; mkdir -p $out
for w in "--beam 16" "--open-lexicon --beam 32" "--open-lexicon --beam 16"; do
  for kn in 0 256 1024 0 1024; do
    timeout -k 10 200 python bench.py --large-utts 0 --no-cpu --also '' --steps 6 --warmup 2 $w --lds-trie $kn > $out/t.json 2> $out/t.err || { tail -3 $out/t.err; exit 1; }
    python - "$w" $kn $out/t.json <<'PY'
import json, sys
d = json.load(open(sys.argv[3])); r = d["roofline"]
print(sys.argv[1], "| staged nodes", sys.argv[2], "| avg launch us %.2f  ms/decode %.3f" % (r["avg_launch_us"], d["ms_per_step"]))
PY
  done
done
