"""Ranks on one GPU: two gloo ranks (decode own shard -> gather_results -> the whole equals a one-rank decode), and
one RCCL rank that drives the final gather through the real all_gather_into_tensor on device buffers."""
import os
import queue
import socket
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

N_TOTAL, BEAM, TMAX, SEED, MARGIN = 301, 8, 24, 99, 60.0


def _decode(lo, hi):
    from silent_speech_amd import synthetic as synth
    from silent_speech_amd.decoder import DeviceTables, NSDecoder
    from silent_speech_amd.tables import NgramCounts
    words = synth.make_lexicon(60, 3)
    corpus = synth.make_sentences(words, 300, 4)
    tg, tl = synth.pack_targets(synth.targets_for_range(words, lo, hi, 2, 3))
    lm = NgramCounts(3, floor_vsz=True).fit(corpus)
    dec = NSDecoder(DeviceTables.from_words(words, lm), BEAM, TMAX, variant="U", lm_weight=0.05, oov_penalty=1.0)
    ids, lens, _ = dec.decode_synth_native(torch.from_numpy(tg).cuda(), torch.from_numpy(tl).cuda(),
                                           base_seed=SEED, margin=MARGIN, utt0=lo)
    torch.cuda.synchronize()
    return ids.clone(), lens.clone()


def _run_ranks(target, world, extra=(), wait_s=300.0):
    """Start `world` fresh processes of `target(rank, world, port, q, *extra)`, return what rank 0 put on the queue.
    A rank that dies early fails the test at once, and whatever still runs afterwards is ended by its own handle."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port, q, *extra)) for r in range(world)]
    try:
        for p in procs:
            p.start()
        t0 = time.monotonic()
        while True:
            try:
                got = q.get(timeout=1.0)
                break
            except queue.Empty:
                bad = [(r, p.exitcode) for r, p in enumerate(procs) if p.exitcode not in (None, 0)]
                assert not bad, f"rank(s) ended before rank 0 reported: {bad}"
                assert time.monotonic() - t0 < wait_s, "no result from rank 0 in time"
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        return got
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
        for p in procs:
            p.join(timeout=10)
            if p.is_alive():
                p.kill()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from silent_speech_amd.dist import gather_results, shard_range
    lo, hi = shard_range(N_TOTAL, rank, world)
    ids, lens = _decode(lo, hi)
    out = gather_results(ids, lens, N_TOTAL, dst=0)
    if rank == 0:
        q.put((out[0].cpu().numpy(), out[1].cpu().numpy()))
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_one_rank():
    g_ids, g_len = _run_ranks(_worker, 2)
    ids, lens = _decode(0, N_TOTAL)
    ids, lens = ids.cpu().numpy(), lens.cpu().numpy()
    assert g_ids.shape == ids.shape and (g_len == lens).all()
    for u in range(N_TOTAL):
        assert (g_ids[u, :lens[u]] == ids[u, :lens[u]]).all(), u
    assert lens.min() >= 1 and len(set(map(bytes, ids))) > N_TOTAL // 2          # different utterances decoded


def _rccl_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from silent_speech_amd.dist import gather_results, last_gathered, unpack_gathered
    assert dist.get_backend() == "nccl"
    ids, lens = _decode(0, N_TOTAL)
    # the deferred form bench.py uses (exchange enqueued, unpacked from the kept receive buffer) ...
    assert gather_results(ids, lens, N_TOTAL, dst=0, unpack=False, force=True) is None
    late = unpack_gathered(last_gathered(N_TOTAL, ids.shape[1], 1, dev), N_TOTAL, ids.shape[1])
    # ... and the direct one, on a second decode's buffers
    out = gather_results(ids, lens, N_TOTAL, dst=0, force=True)
    torch.cuda.synchronize()
    assert out[0].data_ptr() != ids.data_ptr()                 # came out of the receive buffer, not passed through
    assert torch.equal(late[0], out[0]) and torch.equal(late[1], out[1])
    t = torch.tensor([3.5], dtype=torch.float64, device=dev)   # the reductions bench.py runs next to the gather
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    q.put((out[0].cpu().numpy(), out[1].cpu().numpy(), ids.cpu().numpy(), lens.cpu().numpy(), float(t.item())))
    dist.destroy_process_group()


def test_one_rank_rccl_gather_on_device_buffers():
    """The final gather through RCCL (backend "nccl") itself: a process group of one rank on this GPU, the padded
    all_gather_into_tensor on device memory, result equal to the rank's own decode."""
    g_ids, g_len, ids, lens, red = _run_ranks(_rccl_worker, 1)
    assert g_ids.shape == ids.shape and g_ids.dtype == np.uint8
    assert (g_ids == ids).all() and (g_len == lens).all()
    assert red == 3.5
