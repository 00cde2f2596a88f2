"""bench.py's command forms at small sizes: one JSON line each with the contract's fields, the roofline and CPU blocks,
zero transcript mismatches between the GPU and the CPU restatement, and (one launched rank) the process group's
backend in the line."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
SMALL = ["--utts", "256", "--steps", "2", "--warmup", "1", "--timed-decodes", "1", "--cpu-utts", "6", "--cpu-seconds", "20",
         "--cpu-seconds-all", "0"]
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline")


def _line(cmd, env=None, timeout=600):
    p = subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]                   # ONE JSON line on stdout
    return json.loads(lines[0])


def _check_roofline(r):
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.0 < r["frac"] < 1.0
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) < 1e-6 * r["achieved"]


def test_default_line_with_its_blocks():
    d = _line([sys.executable, "bench.py", *SMALL, "--large-utts", "1024", "--shards-in-flight", "2"])
    for k in CONTRACT:
        assert k in d, k
    two = d["two_shards_in_flight"]                # the same utterances as two half-batches on two streams
    assert two["shards"] == 2 and two["equal_to_the_one_stream_decode"] is True and two["value"] > 0
    assert d["unit"] == "chars/s" and d["n_gpus"] == 1 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["config"]["workload"].startswith("configs[1]") and d["value"] > 0 and d["target_chars_per_s"] > 0
    _check_roofline(d["roofline"])
    _check_roofline(d["large_batch"]["roofline"])
    assert d["large_batch"]["targets_recovered_as_prefix"] == 1024
    assert d["variant_U"]["targets_recovered_exactly"] > 200
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["transcript_mismatches_vs_gpu"] == 0


def test_open_lexicon_line():
    d = _line([sys.executable, "bench.py", "--open-lexicon", "--lex-words", "5000", "--beam", "32", "--utts", "128",
               "--steps", "2", "--warmup", "1", "--timed-decodes", "1", "--cpu-utts", "2", "--cpu-seconds", "30",
               "--cpu-seconds-all", "0", "--also="])
    assert d["config"]["workload"].startswith("configs[2]") and "beam 32" in d["metric"]
    _check_roofline(d["roofline"])
    assert d["roofline"]["mean_children_per_live_hyp"] > 8          # a dense lexicon: many children per hypothesis
    assert d["cpu_baseline"]["transcript_mismatches_vs_gpu"] == 0


def test_ctc_rerank_line():
    d = _line([sys.executable, "bench.py", "--ctc-rerank", *SMALL])
    for k in CONTRACT:
        assert k in d, k
    assert d["config"]["workload"].startswith("configs[3]")
    _check_roofline(d["roofline"])
    g = d["ctc_greedy"]
    assert g["kernel"] == "nsd_ctc_greedy_kernel" and g["algorithmic_bytes_per_launch"] >= 256 * 256 * 160
    assert 0 < d["config"]["ctc_transcript_chosen"] <= 256
    c = d["cpu_baseline"]
    assert c["final_transcript_mismatches_vs_gpu"] == 0
    assert c["device_choice_mismatches_vs_restatement_without_postprocess"] == 0


def test_one_launched_rank_goes_through_the_process_group():
    """WORLD_SIZE=1 from a launcher: the nccl (RCCL) process group, barriers, reductions and the final gather run."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    d = _line([sys.executable, "bench.py", "--gpus", "1", "--utts", "256", "--steps", "2", "--warmup", "1",
               "--timed-decodes", "1", "--no-cpu", "--large-utts", "0", "--also="], env=env)
    assert d["config"]["process_group"] == "nccl" and d["config"]["ranks_reported_by_process_group"] == 1
    assert d["config"]["utterances_total"] == 256 and d["value"] > 0
