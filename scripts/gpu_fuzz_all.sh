#!/bin/bash
# Wide randomised parity sweeps on the final kernels (progress goes to files under gpurun_out/).
out=gpurun_out/${1:-fuzz_all}; mkdir -p $out
run() { name=$1; shift; timeout -k 10 1100 python "$@" > $out/$name.log 2>&1; rc=$?; echo "$name rc=$rc: $(tail -1 $out/$name.log)"; [ $rc -eq 0 ] || exit $rc; }
run parity_202 scripts/fuzz_parity.py 1500 202
run parity_404 scripts/fuzz_parity.py 1500 404
run hot_505 scripts/fuzz_parity.py 800 505 -1 hot
NSD_STEP_DENSE=1 run parity_dense1_606 scripts/fuzz_parity.py 800 606
NSD_STEP_DENSE=1 run hot_dense1_707 scripts/fuzz_parity.py 500 707 -1 hot
run dense_a scripts/fuzz_dense.py 11
NSD_STEP_DENSE=1 run dense_b scripts/fuzz_dense.py 12
NSD_STEP_DENSE=0 run dense_c scripts/fuzz_dense.py 13
run dropin scripts/fuzz_dropin.py 300 5
run aux scripts/fuzz_aux.py 30 7
