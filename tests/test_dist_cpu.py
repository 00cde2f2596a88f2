"""world_size-2 gloo test of the sharding + final gather (host logic only)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from silent_speech_amd.dist import gather_results, last_gathered, unpack_gathered, shard_range


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 9, 4096, 1000000):
        for w in (1, 2, 3, 4, 8):
            got = []
            for r in range(w):
                lo, hi = shard_range(n, r, w)
                assert 0 <= lo <= hi <= n
                got.extend(range(lo, hi))
                if n > 1000:
                    got = got[:0]
            if n <= 1000:
                assert got == list(range(n))
            assert sum(shard_range(n, r, w)[1] - shard_range(n, r, w)[0] for r in range(w)) == n


def _worker(rank, world, port, n_total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n_total, rank, world)
    # stand-in for the decode of utterances [lo, hi): ids encode the utterance number
    ids = torch.zeros(hi - lo, 5, dtype=torch.uint8)
    lens = torch.zeros(hi - lo, dtype=torch.int32)
    for i, u in enumerate(range(lo, hi)):
        ids[i] = torch.tensor([37, u % 36, (u // 36) % 36, 38, 0], dtype=torch.uint8)
        lens[i] = 4
    # first the deferred form (exchange only, unpacked from the kept receive buffer), then the direct one
    assert gather_results(ids, lens, n_total, dst=0, unpack=False) is None
    per = (n_total + world - 1) // world
    late = unpack_gathered(last_gathered(per, 5, world, ids.device), n_total, 5) if rank == 0 else None
    out = gather_results(ids, lens, n_total, dst=0)
    if rank == 0:
        assert torch.equal(late[0], out[0]) and torch.equal(late[1], out[1])
        q.put((out[0].numpy(), out[1].numpy()))
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [7, 8, 1])
def test_two_rank_gather_restores_utterance_order(n_total):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    ids, lens = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert ids.shape == (n_total, 5) and (lens == 4).all()
    for u in range(n_total):
        assert list(ids[u, :3]) == [37, u % 36, (u // 36) % 36]
