"""Two ranks on one GPU (gloo): decode own shard -> gather_results -> the whole thing equals a one-rank decode."""
import os
import socket

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
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    g_ids, g_len = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ids, lens = _decode(0, N_TOTAL)
    ids, lens = ids.cpu().numpy(), lens.cpu().numpy()
    assert g_ids.shape == ids.shape and (g_len == lens).all()
    for u in range(N_TOTAL):
        assert (g_ids[u, :lens[u]] == ids[u, :lens[u]]).all(), u
    assert lens.min() >= 1 and len(set(map(bytes, ids))) > N_TOTAL // 2          # different utterances decoded
