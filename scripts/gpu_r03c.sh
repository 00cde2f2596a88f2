#!/bin/bash
# Round-3 evidence: reference fixtures on the GPU, cold / warm phase times of a wave, rocprofv3 statistics of the
# CTC + rerank command, then the default command's bench line, kernel statistics and PMC passes.
out=gpurun_out/r03c; mkdir -p $out
root=$PWD
timeout -k 10 400 python -u -m pytest tests/test_gpu_reference_tail.py -m gpu -x -q -s --timeout 300 --timeout-method thread > $out/pytest_tail.log 2>&1; rc=$?
tail -2 $out/pytest_tail.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python scripts/phase_repeat.py I 16 64 4096 > $out/phase_repeat_I.txt 2>&1 || { tail -5 $out/phase_repeat_I.txt; exit 1; }
timeout -k 10 200 python scripts/phase_repeat.py U 16 64 4096 > $out/phase_repeat_U.txt 2>&1 || { tail -5 $out/phase_repeat_U.txt; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$root/$out/ctc_stats" -o s -- python3 "$root/bench.py" --ctc-rerank --no-cpu > "$root/$out/ctc_stats.log" 2>&1 || { tail -5 "$root/$out/ctc_stats.log"; exit 1; }
rm -f "$root/$out"/ctc_stats/*kernel_trace.csv
cd "$root"
bash scripts/refresh_profiles.sh r03c_default
