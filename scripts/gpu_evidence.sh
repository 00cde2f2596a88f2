#!/bin/bash
# Measurement evidence in one GPU call: the whole GPU test suite, the default command's bench line / rocprofv3 kernel
# statistics / PMC passes, the other bench shapes with their kernel statistics, traffic and busy counters, the CTC
# kernel beyond the Infinity Cache, instructions per phase.  Outputs under gpurun_out/<tag>_*
# (usage: gpu_evidence.sh [tag]).
set -o pipefail
root=$PWD
tag=${1:-ev}
out=$root/gpurun_out/${tag}_final; mkdir -p $out
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 900 --timeout-method thread > $out/pytest_gpu.log 2>&1; rc=$?
tail -2 $out/pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
bash scripts/refresh_profiles.sh ${tag}_default || exit 1
stats() {   # stats <name> <bench flags...>
  name=$1; shift
  ( cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_$name" -o s -- python3 "$root/bench.py" "$@" > "$out/stats_$name.log" 2>&1 ) || { tail -5 "$out/stats_$name.log"; return 1; }
  rm -f "$out"/stats_$name/*kernel_trace.csv
}
timeout -k 10 300 python bench.py --no-cpu --utts 131072 --steps 5 --warmup 1 --also= > $out/bench_big.json 2> $out/bench_big.err || exit 1
timeout -k 10 300 python bench.py --open-lexicon --beam 32 --no-cpu --steps 6 --also= > $out/bench_open32.json 2> $out/bench_open32.err || exit 1
timeout -k 10 400 python bench.py --ctc-rerank > $out/bench_ctc.json 2> $out/bench_ctc.err || exit 1
stats big --no-cpu --utts 131072 --steps 5 --warmup 1 --also= || exit 1
stats open32 --open-lexicon --beam 32 --no-cpu --steps 6 --also= || exit 1
stats ctc --ctc-rerank --no-cpu || exit 1
( cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_ctc32768" -o s -- python3 "$root/scripts/bench_ctc.py" 32768 256 > "$out/ctc32768.log" 2>&1 ) || { tail -5 $out/ctc32768.log; exit 1; }
rm -f "$out"/stats_ctc32768/*kernel_trace.csv
bash scripts/pmc_traffic.sh ${tag}_traffic_big "131072 utterances, beam 16, variant I" --utts 131072 > $out/traffic_big.log 2>&1 || { tail -5 $out/traffic_big.log; exit 1; }
bash scripts/pmc_traffic.sh ${tag}_traffic_open "configs[2]: 4096 utterances, beam 32, 100k-word lexicon, variant I" --open-lexicon --beam 32 > $out/traffic_open.log 2>&1 || { tail -5 $out/traffic_open.log; exit 1; }
bash scripts/pmc_busy.sh ${tag}_busy_4096 > $out/busy_4096.log 2>&1 || { tail -5 $out/busy_4096.log; exit 1; }
bash scripts/pmc_busy.sh ${tag}_busy_131072 --utts 131072 > $out/busy_big.log 2>&1 || { tail -5 $out/busy_big.log; exit 1; }
if [ -f build/diag/libnsd_stop4.so ]; then
  bash scripts/phase_insts.sh ${tag}_insts_4096 I 4096 16 20 > $out/insts_4096.txt 2>&1 || { tail -5 $out/insts_4096.txt; exit 1; }
  bash scripts/phase_insts.sh ${tag}_insts_131072 I 131072 16 20 > $out/insts_131072.txt 2>&1 || { tail -5 $out/insts_131072.txt; exit 1; }
  bash scripts/phase_insts.sh ${tag}_insts_open32 I 4096 32 20 open > $out/insts_open32.txt 2>&1 || { tail -5 $out/insts_open32.txt; exit 1; }
fi
echo evidence done
