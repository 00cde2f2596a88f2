#!/bin/bash
# One GPU call for kernel experiments: parity of the in-tree library (pytest subset + fuzz), then the default bench of
# each library given on the command line (side by side, twice), then the large batch of each.
# Usage: scripts/gpu_exp.sh <tag> <fuzz cases> lib1.so lib2.so ...
tag=$1; nf=$2; shift 2
out=gpurun_out/$tag; mkdir -p $out
if [ "$nf" -gt 0 ]; then
  timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 --timeout-method thread > $out/pytest.log 2>&1; rc=$?
  echo "pytest rc=$rc"; tail -2 $out/pytest.log
  [ $rc -ne 0 ] && exit $rc
  timeout -k 10 400 python scripts/fuzz_parity.py $nf 303 > $out/fuzz.log 2>&1; rc=$?
  echo "fuzz rc=$rc"; tail -1 $out/fuzz.log
  [ $rc -ne 0 ] && exit $rc
fi
bash scripts/bench_libs.sh "$@" || exit 1
for arg in "$@"; do
  so=${arg%%:*}; mode=${arg#*:}; [ "$mode" = "$arg" ] && mode=9
  t=$(basename $so)_m$mode
  export NSD_STEP_MODE=$mode
  NSDECODE_LIB=$PWD/$so timeout -k 10 200 python bench.py --large-utts 0 --no-cpu --utts 131072 --steps 5 --warmup 1 --also= > $out/big_$t.json 2> $out/big_$t.err || exit 1
  NSDECODE_LIB=$PWD/$so timeout -k 10 200 python bench.py --large-utts 0 --open-lexicon --beam 32 --no-cpu --steps 6 --also= > $out/open_$t.json 2> $out/open_$t.err || exit 1
  python - $arg $out/big_$t.json $out/open_$t.json <<'PY'
import json, sys
b = json.load(open(sys.argv[2]))["roofline"]; o = json.load(open(sys.argv[3]))["roofline"]
print(sys.argv[1], "big launch %.1f us frac %.3f | open32 launch %.2f us frac %.3f" % (b["avg_launch_us"], b["frac"], o["avg_launch_us"], o["frac"]))
PY
done
