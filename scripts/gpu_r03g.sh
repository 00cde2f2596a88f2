#!/bin/bash
# A/B of nsd_step builds (parity of the in-tree one first), then phase stamps on the open lexicon at beam 32.
out=gpurun_out/r03g; mkdir -p $out
bash scripts/gpu_exp.sh r03g 600 "$@" || exit 1
timeout -k 10 300 python scripts/phase_profile.py I 4096 32 open > $out/stamps_open32.txt 2>&1 || { tail -5 $out/stamps_open32.txt; exit 1; }
grep "^step" $out/stamps_open32.txt | cut -c1-250
