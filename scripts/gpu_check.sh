#!/bin/bash
# One GPU call: parity suite, a fuzz sweep, then the three benchmark shapes.  Usage: scripts/gpu_check.sh <tag> [fuzz cases]
tag=$1; nf=${2:-600}
out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $out/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -2 $out/pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python scripts/fuzz_parity.py $nf 303 > $out/fuzz.log 2>&1; rc=$?
echo "fuzz rc=$rc"; tail -1 $out/fuzz.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python bench.py --large-utts 0 --no-cpu > $out/default.json 2> $out/default.err || exit 1
timeout -k 10 200 python bench.py --large-utts 0 --open-lexicon --beam 32 --no-cpu --steps 6 > $out/open32.json 2> $out/open32.err || exit 1
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
