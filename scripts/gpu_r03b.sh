#!/bin/bash
# Round-3 check: the new GPU tests, the default bench line with its large_batch block, the CTC + rerank workload,
# the driver's one-rank torchrun line, and variant U per instantiation mode.
out=gpurun_out/r03b; mkdir -p $out
timeout -k 10 300 python -u -m pytest tests/test_dist_gpu.py tests/test_gpu_parity.py -m gpu -x -q --timeout 200 --timeout-method thread -k "rank or refit or rejects or same_size" > $out/pytest.log 2>&1; rc=$?
tail -3 $out/pytest.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
timeout -k 10 400 python bench.py --ctc-rerank > $out/ctc.json 2> $out/ctc.err || { tail -5 $out/ctc.err; exit 1; }
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu --large-utts 0 > $out/torchrun1.json 2> $out/torchrun1.err || { tail -5 $out/torchrun1.err; exit 1; }
for m in 0 1 9; do
  NSD_STEP_MODE=$m timeout -k 10 200 python bench.py --variant U --also= --no-cpu --large-utts 0 > $out/U_m$m.json 2> $out/U_m$m.err || exit 1
done
python - <<'PY'
import json
for f in ("bench", "ctc", "torchrun1", "U_m0", "U_m1", "U_m9"):
    d = json.loads(open(f"gpurun_out/r03b/{f}.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f, "%.1f Mchars/s ms/step %.3f launch %.2f us frac %.3f" % (d["value"] / 1e6, d["ms_per_step"], r["avg_launch_us"], r["frac"]))
PY
