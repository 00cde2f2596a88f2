#!/bin/bash
# Both emission forms of nsd_step forced in turn (parity), then the A/B of libraries given on the command line.
out=gpurun_out/r03h; mkdir -p $out
timeout -k 10 900 python -u -m pytest tests/test_gpu_dense_forms.py -m gpu -x -q --timeout 900 --timeout-method thread > $out/pytest_dense.log 2>&1; rc=$?
tail -3 $out/pytest_dense.log; [ $rc -ne 0 ] && exit $rc
bash scripts/gpu_exp.sh r03h 300 "$@"
