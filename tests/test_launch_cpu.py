"""The self-starting multi-rank path of bench.py, rehearsed on the CPU with gloo and a stub worker."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]

STUB = r"""
import json, os, sys
import torch
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["LOCAL_RANK"]) == rank
dist.init_process_group("gloo")
t = torch.tensor([float(rank + 1)])
dist.all_reduce(t)
if "--fail-on" in sys.argv and rank == int(sys.argv[sys.argv.index("--fail-on") + 1]):
    sys.exit(7)
print(json.dumps({"rank": rank, "n_gpus": dist.get_world_size(), "sum": t.item()}), flush=True)
dist.barrier()
dist.destroy_process_group()
"""

DRIVER = r"""
import sys
sys.path.insert(0, {root!r})
from silent_speech_amd.launch import spawn_ranks
sys.exit(spawn_ranks([sys.executable, "-c", {stub!r}, *sys.argv[2:]], int(sys.argv[1]), timeout=120))
"""


def _run(world, *extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    return subprocess.run([sys.executable, "-c", DRIVER.format(root=str(ROOT), stub=STUB), str(world), *extra],
                          capture_output=True, text=True, timeout=300, env=env)


def test_launcher_starts_the_ranks_and_relays_rank_zero():
    p = _run(2)
    assert p.returncode == 0, p.stderr
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert lines == [{"rank": 0, "n_gpus": 2, "sum": 3.0}]           # only rank 0 reaches stdout
    assert '"rank": 1' in p.stderr


def test_launcher_ends_the_other_ranks_when_one_fails():
    p = _run(3, "--fail-on", "1")
    assert p.returncode == 7
    assert p.stdout.strip() == "" or '"rank": 0' in p.stdout


def test_bench_gpus_n_starts_n_ranks_itself():
    """`python bench.py --gpus 2` without WORLD_SIZE: the parent starts two fresh ranks (here each one stops at
    once because no GPU is visible) and reports failure instead of measuring one GPU as if it were two."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""                                   # also on a GPU box: no device for the ranks
    env["CUDA_VISIBLE_DEVICES"] = ""
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--no-cpu", "--steps", "1"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode != 0
    # (the launcher ends the other rank as soon as one has failed: the second message may not get out)
    assert 1 <= p.stderr.count("no HIP device visible") <= 2
    assert '"n_gpus"' not in p.stdout


def test_bench_worker_rejects_a_world_size_that_differs_from_gpus():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--no-cpu"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode != 0 and "WORLD_SIZE=4" in p.stderr
