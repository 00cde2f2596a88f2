#!/bin/bash
# Round-3 diagnostics: phase stamps on the open lexicon at beam 32 and on variant U, the large batch with one and two
# utterances per wave, traffic counters of the large batch and of the open lexicon.
out=gpurun_out/r03e; mkdir -p $out
timeout -k 10 300 python scripts/phase_profile.py I 4096 32 open > $out/stamps_open32.txt 2>&1 || { tail -5 $out/stamps_open32.txt; exit 1; }
timeout -k 10 300 python scripts/phase_profile.py U 4096 16 > $out/stamps_U.txt 2>&1 || { tail -5 $out/stamps_U.txt; exit 1; }
bash scripts/gpu_big.sh r03e 9:32 9:64 1:32 || exit 1
bash scripts/pmc_traffic.sh r03e_traffic_big "131072 utterances, beam 16, variant I" --utts 131072 > $out/traffic_big.log 2>&1 || { tail -5 $out/traffic_big.log; exit 1; }
bash scripts/pmc_traffic.sh r03e_traffic_open "configs[2]: 4096 utterances, beam 32, 100k-word lexicon, variant I" --open-lexicon --beam 32 > $out/traffic_open.log 2>&1 || { tail -5 $out/traffic_open.log; exit 1; }
tail -2 $out/traffic_big.log $out/traffic_open.log
