#!/bin/bash
# Round-3 check of the wave-per-string LM kernels and the new argument checks, then launch time against batch size.
out=gpurun_out/r03d; mkdir -p $out
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_tail.py tests/test_zz_no_alloc.py tests/test_ar_head.py -m gpu -x -q --timeout 300 --timeout-method thread -k "lm_ or rerank or ctc or refused or refuse or no_alloc or alloc or ar_ or joint or tail" > $out/pytest.log 2>&1; rc=$?
tail -3 $out/pytest.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python scripts/fuzz_aux.py 30 11 > $out/fuzz_aux.log 2>&1; rc=$?; tail -2 $out/fuzz_aux.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --ctc-rerank --no-cpu > $out/ctc.json 2> $out/ctc.err || { tail -5 $out/ctc.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03d/ctc.json").read().strip().splitlines()[-1])
print("ctc-rerank %.1f Mchars/s ms/step %.3f" % (d["value"] / 1e6, d["ms_per_step"]), d["ctc_greedy"]["avg_launch_us"], d["rerank_lm"], d["share_of_step"])
PY
timeout -k 10 300 python scripts/launch_vs_n.py I 16 1 64 1024 4096 > $out/launch_vs_n_I.txt 2>&1 || exit 1
timeout -k 10 300 python scripts/launch_vs_n.py U 16 1 64 1024 4096 > $out/launch_vs_n_U.txt 2>&1 || exit 1
cat $out/launch_vs_n_I.txt $out/launch_vs_n_U.txt
